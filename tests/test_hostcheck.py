"""The HOST half of libmspack_amd/csrc/hip/shim.hip under real AddressSanitizer + UBSan and under real ThreadSanitizer (VERDICT round 5,
item 1a: the chunk planner, the page-lock registry and the cut copies, the copy-back thread, the shard threads and the staging pool
had never run under a sanitizer -- the wavefront emulator only borrows TSan's hooks).  tests/hostcheck builds shim.hip for the host
with -DMSPACK_HOST_CHECK: streams are real queues with worker threads, a launch's place in its stream is taken by the CPU stand-in
(one oracle call per unit), and the page-lock rules the HIP runtime was found to have are modelled and counted:
overlapping registrations, unregistering what is not registered, copies that straddle a registration's boundary.  A scenario passes
when its outputs equal the plaintext, no modelled rule was violated and the sanitizer has nothing to say.  CPU only."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "hostcheck", "build_hostcheck.sh")
FOUR_CHUNKS = {"MSPACK_HIP_CHUNK_BYTES": "65536", "MSPACK_HIP_CHUNK_UNITS": "16"}     # (the copy-back thread only runs for several chunks)
SCENARIOS = [("partial_pins", ["2"], {}), ("partial_pins", ["2"], FOUR_CHUNKS), ("ownership", [], {}), ("ownership", [], FOUR_CHUNKS),
             ("shards_threads", [], FOUR_CHUNKS), ("lifetimes", ["3"], {}), ("lifetimes", ["1"], FOUR_CHUNKS),
             ("jobs", ["3"], FOUR_CHUNKS), ("jobs", ["1"], {})]


@pytest.fixture(scope="module")
def hostcheck_bins():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("no clang with sanitizer runtimes")
    ps = [subprocess.Popen(["bash", BUILD, k], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for k in ("asan", "tsan")]
    for p in ps:
        out = p.communicate()[0].decode(errors="replace")
        assert p.returncode == 0, out[-4000:]
    return {k: os.path.join(ROOT, "tests", "_build", "hostcheck_" + k) for k in ("asan", "tsan")}


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_shim_host_half_under_sanitizers(hostcheck_bins, san):
    env0 = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:exitcode=97", UBSAN_OPTIONS="print_stacktrace=1",
                TSAN_OPTIONS="halt_on_error=0:exitcode=66")
    running = []
    for name, args, env in SCENARIOS:
        running.append((name, env, subprocess.Popen([hostcheck_bins[san], name] + args, env=dict(env0, **env), stdout=subprocess.PIPE,
                                                    stderr=subprocess.STDOUT)))
    for name, env, p in running:
        out = p.communicate(timeout=900)[0].decode(errors="replace")
        assert p.returncode == 0 and ("HOSTCHECK_OK " + name) in out, (san, name, env, out[-4000:])
        assert "Sanitizer" not in out and "runtime error" not in out and "VIOLATION" not in out, (san, name, env, out[-4000:])
