"""The object API driven from C (libmspack_amd/csrc/bench/api_bench.c: an in-memory mspack_system, open + extract of every file)
on BASELINE configs 3 and 4 as CONTAINERS -- one CHM of 1024 LZX reset intervals (window 2^21, reset every 2 frames, 61 files),
one cabinet of 512 Quantum folders (the config-2 cabinet has its own test, tests/test_config2_cab.py) -- the path bench.py's
`through_api` times.
  * CPU, host logic: the harness + the C drivers on the oracle-backed stand-in for the batch ABI, small containers;
  * CPU, where oracle/_ref exists: the REAL reference extracts the same containers to the plaintext (the corpus is pinned);
  * GPU: the full-size containers through libmspack_hip.so, every extracted byte against the plaintext."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

import libmspack_amd as M
from libmspack_amd import apibench
import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness_cpu(built):
    """api_bench.c + the host drivers + the stand-in for the batch ABI in one CPU library (test infrastructure)"""
    bdir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(bdir, exist_ok=True)
    so = os.path.join(bdir, "libapibench_cpu.so")
    srcs = [os.path.join(ROOT, "libmspack_amd", "csrc", "bench", "api_bench.c")] + \
        sorted(glob.glob(os.path.join(ROOT, "libmspack_amd", "csrc", "host", "*.c"))) + \
        [os.path.join(ROOT, "tests", "csrc", "batch_standin.c")] + sorted(glob.glob(os.path.join(ROOT, "oracle", "*_oracle.c")))
    deps = srcs + glob.glob(os.path.join(ROOT, "include", "*.h"))
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in deps):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include"),
                               "-o", so] + srcs + ["-lpthread"])
    L = C.CDLL(so)
    for fn in (L.mspk_api_bench_cab, L.mspk_api_bench_chm):
        fn.restype = C.c_int
        fn.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint, C.POINTER(apibench.Stats)]
    return L


def run_with(L, kind, image, out_cap):
    img = np.frombuffer(bytes(image), dtype=np.uint8)
    out = np.zeros(out_cap + 64, dtype=np.uint8)
    offs = np.zeros(70000, dtype=np.uint64)
    st = apibench.Stats()
    fn = L.mspk_api_bench_cab if kind == "cab" else L.mspk_api_bench_chm
    rc = fn(img.ctypes.data, img.size, out.ctypes.data, out_cap, offs.ctypes.data, 70000, C.byref(st))
    return rc, out[:st.bytes_out], offs[:st.n_files + 1], st


def check_chm(out, offs, st, plain, slices):
    # the directory lists the files in name order = the order they were laid out in
    assert st.n_errors == 0 and st.n_files == len(slices) and st.bytes_out == plain.size
    for k, (o, l) in enumerate(slices):
        assert np.array_equal(out[int(offs[k]):int(offs[k + 1])], plain[o:o + l]), k


def test_harness_host_logic_cpu(harness_cpu):
    cab, plain = apibench.build_config2_cab(M, n=24)
    rc, out, offs, st = run_with(harness_cpu, "cab", cab, plain.size)
    assert rc == 0 and st.n_errors == 0 and st.n_files == 24 and np.array_equal(out, plain)
    chm, plain, slices = apibench.build_config3_chm(M, n=12, n_files=7)
    rc, out, offs, st = run_with(harness_cpu, "chm", chm, plain.size)
    assert rc == 0
    check_chm(out, offs, st, plain, slices)
    cab, plain = apibench.build_config4_cab(M, n=3, frames=4, window_bits=18)
    rc, out, offs, st = run_with(harness_cpu, "cab", cab, plain.size)
    assert rc == 0 and st.n_errors == 0 and st.n_files == 3 and np.array_equal(out, plain)


@pytest.mark.skipif(not helpers.have_ref(), reason="oracle/_ref (the compiled reference) is not present")
def test_containers_reference_cpu(built):
    """the real libmspack extracts the (smaller) containers of the same recipes to the plaintext"""
    chm, plain, slices = apibench.build_config3_chm(M, n=64, n_files=9)
    e, lst = helpers.ref_chm_list(chm)
    assert e == 0 and len(lst) == len(slices)
    rc, got = helpers.ref_chm_extract(chm, list(range(len(slices))))
    assert rc == 0
    # (the real chmd lists the files in directory order = name order = layout order)
    for k, (o, l) in enumerate(slices):
        assert got[k][0] == 0 and got[k][1] == plain[o:o + l].tobytes(), k
    cab, plain = apibench.build_config4_cab(M, n=6, frames=8, window_bits=21)
    rc, got = helpers.ref_cab_extract(cab, list(range(6)), cap=plain.size + 4096)
    assert rc == 0
    for i, (err, data) in enumerate(got):
        assert err == 0 and data == plain[i * 8 * 32768:(i + 1) * 8 * 32768].tobytes(), i


@pytest.mark.gpu
def test_config3_chm_gpu(built):
    chm, plain, slices = apibench.build_config3_chm(M)
    rc, out, offs, d = apibench.run("chm", chm, plain.size)
    assert rc == 0 and d["n_errors"] == 0 and d["n_files"] == len(slices) and d["bytes_out"] == plain.size
    for k, (o, l) in enumerate(slices):
        assert np.array_equal(out[int(offs[k]):int(offs[k + 1])], plain[o:o + l]), k
    print(apibench.summary(d))


@pytest.mark.gpu
def test_config4_quantum_cab_gpu(built):
    cab, plain = apibench.build_config4_cab(M)
    rc, out, offs, d = apibench.run("cab", cab, plain.size)
    assert rc == 0 and d["n_errors"] == 0 and d["n_files"] == 512 and np.array_equal(out, plain)
    print(apibench.summary(d))


@pytest.mark.gpu
def test_config2_cab_through_c_harness_gpu(built):
    cab, plain = apibench.build_config2_cab(M)
    rc, out, offs, d = apibench.run("cab", cab, plain.size)
    assert rc == 0 and d["n_errors"] == 0 and d["n_files"] == 4096 and np.array_equal(out, plain)
    print(apibench.summary(d))
