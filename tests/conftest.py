import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _abort_backtrace()


def _abort_backtrace():
    """If anything in this process calls abort() -- the HIP runtime on a GPU memory fault, glibc on a corrupted heap, an assertion --
    say WHO: tools/csrc/abort_bt.c prints the native backtrace of the aborting thread to the terminal pytest was started on (its
    own dup of stderr, taken here, before any capture) and to tests/_build/abort_bt.log.  Round 5 lost a whole-suite run to a SIGABRT
    whose message went into pytest's capture and died with the process."""
    import ctypes
    import subprocess
    try:
        bdir = os.path.join(ROOT, "tests", "_build")
        os.makedirs(bdir, exist_ok=True)
        so = os.path.join(bdir, "abort_bt.so")
        src = os.path.join(ROOT, "tools", "csrc", "abort_bt.c")
        if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
            subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, src])
        os.environ.setdefault("ABORT_BT_LOG", os.path.join(bdir, "abort_bt.log"))
        ctypes.CDLL(so)
    except Exception:            # (diagnostics only: never a reason to fail a run)
        pass


# Collection order: the driver runs `pytest -m gpu -x`, so a failure in a container / driver test
# must not hide the codec-level oracle parity behind it.  Codec parity first (known-answer vectors, then LZX, MSZIP, Quantum
# against the oracle), then the host-buffer path and the fuzz sweeps, then the object API's drivers and containers, the
# slow shapes last.  Within a module the order is the file's.
_ORDER = ["test_gpu_kat", "test_gpu_lzx", "test_gpu_lzx_frames", "test_gpu_lzx_log", "test_gpu_lzxd", "test_gpu_mszip",
          "test_gpu_mszip_blocks", "test_gpu_fold", "test_gpu_qtm", "test_szdd_kwaj", "test_oab", "test_gpu_hostpath", "test_gpu_fuzz",
          "test_gpu_messages", "test_gpu_drivers", "test_chm_extract", "test_chmdir", "test_chm_messages", "test_cab_sticky",
          "test_cabsets", "test_config2_cab", "test_gpu_reference_suites", "test_api_bench", "test_gpu_bench_line",
          "test_gpu_large_files"]


def pytest_collection_modifyitems(session, config, items):
    rank = {m: i for i, m in enumerate(_ORDER)}

    def key(it):
        mod = os.path.splitext(os.path.basename(str(it.fspath)))[0]
        return rank.get(mod, len(_ORDER) // 2)
    items.sort(key=key)           # (stable: a module's tests keep their order)
    # The one test that has ever taken the whole process down (round 5: SIGABRT inside the HIP runtime, once in five whole-suite
    # runs, never again in round 6's eight) runs LAST: if it ever does that again, every other result is
    # already on the terminal -- and the native backtrace follows (_abort_backtrace above).
    last = [it for it in items if it.name.startswith("test_copies_are_cut_at_pin_boundaries")]
    if last:
        items[:] = [it for it in items if it not in last] + last


@pytest.fixture(scope="session")
def built():
    """Build (or reuse) the native libraries once per session."""
    from libmspack_amd import build
    build.build_corpus()
    build.build_oracle()
    build.build_hip()
    return True


@pytest.fixture(scope="session")
def hostlogic(built):
    """tests/_build/libhostlogic_cpu.so: the host drivers' C files (libmspack_amd/csrc/host/*.c) linked with
    tests/csrc/batch_standin.c, a CPU stand-in for the batch ABI built on the oracle -- so that the HOST logic
    behind include/mspack.h can be tested without a GPU.  Test infrastructure only; the product library never
    contains or loads it."""
    import ctypes
    import glob
    import subprocess
    bdir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(bdir, exist_ok=True)
    so = os.path.join(bdir, "libhostlogic_cpu.so")
    srcs = sorted(glob.glob(os.path.join(ROOT, "libmspack_amd", "csrc", "host", "*.c"))) + \
        [os.path.join(ROOT, "tests", "csrc", "batch_standin.c")] + \
        sorted(glob.glob(os.path.join(ROOT, "oracle", "*_oracle.c")))
    deps = srcs + glob.glob(os.path.join(ROOT, "include", "*.h")) + glob.glob(os.path.join(ROOT, "oracle", "*.h"))
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in deps):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
                               "-I", os.path.join(ROOT, "include"), "-o", so] + srcs + ["-lpthread"])
    return ctypes.CDLL(so)
