"""chmd->extract() pinned to the REAL reference chmd.

tests/golden/chm_extract.json holds, for 22 synthetic CHMs (tests/chm_extract_recipe.py rebuilds them
byte for byte from the recipe; the golden keeps their MD5), what the reference's chmd answered for every
extract() call of several call orders on ONE decompressor: error code, bytes written, MD5 of those bytes
(made by tests/golden/make_chm_extract_golden.py with oracle/_ref in the development container).  The
calls are replayed through include/mspack.h with the same in-memory mspack_system semantics
(api.MemSystem) -- cf. libmspack/test/chmd_order.c:55-129, chmd.c:906-1041,1072-1315.

  * `-m gpu`: against libmspack_hip.so, i.e. the HIP kernels (the parity test proper);
  * `-m "not gpu"`: the same host driver code (csrc/host/chmd.c) linked with the CPU stand-in for the
    batch ABI (tests/csrc/batch_standin.c) -- host logic only."""
import hashlib
import json
import os

import pytest

from libmspack_amd import api
import chm_extract_recipe as R

VECS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "chm_extract.json")))


def replay(v, L=None):
    chm, _d, files = R.build(v["case"])
    assert hashlib.md5(chm).hexdigest() == v["chm_md5"], "recipe no longer reproduces the golden CHM"
    for run in v["runs"] or [None]:
        with api.Chm(chm, mem=True, L=L) as c:
            assert c.open_error == v["open_err"], v["tag"]
            if run is None:
                continue
            assert [(n, off, ln) for n, ln, off, _s in c.files] == files
            for k, (idx, exp) in enumerate(zip(run["order"], run["results"])):
                err, data = c.extract(idx)
                tag = "%s order %s call %d (file %d)" % (v["tag"], run["order"], k, idx)
                if err != exp["err"]:       # what the driver said, and the batch ABI's own error text, belong in the report
                    said = list(c.mem.messages) if c.mem else []
                    hip = (L or api._setup(None)).mspack_hip_last_error
                    hip.restype = __import__("ctypes").c_char_p
                    raise AssertionError((tag, err, exp, said, hip()))
                assert len(data) == exp["n"], (tag, len(data), exp)
                assert hashlib.md5(data).hexdigest() == exp["md5"], tag


@pytest.mark.gpu
@pytest.mark.parametrize("v", VECS, ids=[v["tag"] for v in VECS])
def test_chm_extract_vs_reference_gpu(built, v):
    replay(v)


@pytest.mark.parametrize("v", VECS, ids=[v["tag"] for v in VECS])      # (config 3's 64 MiB CHM too: ~20 s on the stand-in)
def test_chm_extract_host_logic_cpu(built, hostlogic, v):
    replay(v, L=hostlogic)


def test_chm_big_reset_interval_small_table(built, hostlogic):
    """ADVICE round 2 (chmd.c setup_sec1): ControlData may state a reset interval of up to 65535 frames while the reset
    table has a handful of entries; the per-frame offset table the driver builds behind the compressed stream then
    holds one interval's worth of entries and must have room for them.  Same answers as the real chmd (where the
    reference is built), no crash anywhere."""
    import helpers
    base = dict(seed=77, text=0, n_bytes=200000, window_bits=16, reset_frames=2, files=R.spread_files(200000, 5, 3, 65536))
    for frames in (1024, 60000, 65535):
        case = dict(base, mutations=[["control_u32", 0x0C, frames]])      # version 2: counted in 32 KiB frames
        chm, _d, files = R.build(case)
        want = None
        if helpers.have_ref():
            rc, want = helpers.ref_chm_extract(chm, list(range(len(files))))
            want = None if rc else want
        with api.Chm(chm, mem=True, L=hostlogic) as c:
            if c.open_error:
                continue
            for i in range(len(files)):
                err, data = c.extract(i)
                if want is not None:
                    assert err == want[i][0] and (err or data == want[i][1]), (frames, i, err, want[i][0])


def test_chm_chunks_run_as_jobs(built, hostlogic):
    """A chunk's batch runs as a job (include/mspack_hip.h: mspack_hip_decode_batch_begin): extract() of file i waits for the
    intervals file i needs, not for the chunk.  The stand-in's job is lazy and hostile (tests/csrc/batch_standin.c: results
    poisoned at _begin, a unit decoded only when it is waited for), so a driver that read a result or a byte it had not waited
    for would fail the goldens above; here: the jobs ARE taken (every multi-interval golden went through one), and the
    synchronous way (MSPACK_HIP_JOBS=0, read per call by the stand-in) gives the same bytes file by file, in several orders."""
    import ctypes
    hostlogic.mspack_standin_jobs_begun.restype = ctypes.c_ulong
    hostlogic.mspack_standin_job_waits.restype = ctypes.c_ulong
    v = next(x for x in VECS if x["tag"].startswith("config3") or len(x["case"].get("files", [])) >= 8)
    chm, _d, files = R.build(v["case"])
    orders = [list(range(len(files))), list(reversed(range(len(files)))), [len(files) // 2, 0, len(files) - 1, 1]]
    got = {}
    for jobs in ("1", "0"):
        os.environ["MSPACK_HIP_JOBS"] = jobs
        try:
            b0, w0 = hostlogic.mspack_standin_jobs_begun(), hostlogic.mspack_standin_job_waits()
            for oi, order in enumerate(orders):
                with api.Chm(chm, mem=True, L=hostlogic) as c:
                    assert not c.open_error
                    for i in order:
                        err, data = c.extract(i)
                        got.setdefault((oi, i), []).append((err, hashlib.md5(data).hexdigest(), len(data)))
            b1, w1 = hostlogic.mspack_standin_jobs_begun(), hostlogic.mspack_standin_job_waits()
            if jobs == "1":
                assert b1 > b0 and w1 > w0, "the CHM driver did not take the job path"
            else:
                assert b1 == b0, "MSPACK_HIP_JOBS=0 still began a job"
        finally:
            os.environ.pop("MSPACK_HIP_JOBS", None)
    for key, pair in got.items():
        assert pair[0] == pair[1], (key, pair)
        assert pair[0][0] == 0
