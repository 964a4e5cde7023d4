"""CPU: the C-ABI library loads and exports every function that include/*.h declares, its structs
have the layout the Python mirror assumes, and the product fails loudly without a GPU (no CPU
fallback path exists).  No compute calls are made here."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

import libmspack_amd as M
from libmspack_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(mspack_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if not n.endswith("_internal") or True))


def test_every_declared_symbol_is_exported(built):
    L = M.lib()
    for hdr in ("mspack_hip.h", "mspack.h"):
        for name in declared_functions(hdr):
            assert hasattr(L, name), "%s (declared in %s) is not exported" % (name, hdr)
    for name in M.EXPORTED_SYMBOLS:
        assert hasattr(L, name)


def test_struct_layouts_match_the_c_headers(built):
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "mspack_hip.h"
#include "mspack.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(mspack_hip_unit), sizeof(mspack_hip_result),
         offsetof(mspack_hip_unit, kind), offsetof(mspack_hip_unit, flags), offsetof(mspack_hip_result, good_len),
         sizeof(struct mspack_system));
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(struct mscabd_cabinet), sizeof(struct mscabd_folder),
         sizeof(struct mscabd_file), sizeof(struct mschmd_header), sizeof(struct mschmd_file),
         sizeof(struct mscab_decompressor));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "t.c")
        open(src, "w").write(prog)
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-D_FILE_OFFSET_BITS=64", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        a, b = [list(map(int, ln.split())) for ln in subprocess.check_output([exe]).decode().splitlines()]
    assert a[0] == M.UNIT_DTYPE.itemsize and a[1] == M.RESULT_DTYPE.itemsize
    assert a[2] == M.UNIT_DTYPE.fields["kind"][1] and a[3] == M.UNIT_DTYPE.fields["flags"][1]
    assert a[4] == M.RESULT_DTYPE.fields["good_len"][1]
    assert a[5] == 11 * 8
    assert b == [C.sizeof(api.MscabdCabinet), C.sizeof(api.MscabdFolder), C.sizeof(api.MscabdFile),
                 C.sizeof(api.MschmdHeader), C.sizeof(api.MschmdFile), C.sizeof(api.MscabDecompressor)]


def test_reference_layout_compat(built):
    """When the reference headers are around (dev container), our structs are byte-compatible."""
    ref = "/root/reference/libmspack/mspack/mspack.h"
    if not os.path.exists(ref):
        pytest.skip("reference headers absent")
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include <mspack.h>
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(struct mscabd_cabinet), sizeof(struct mscabd_folder),
         sizeof(struct mscabd_file), sizeof(struct mschmd_header), sizeof(struct mschmd_file),
         sizeof(struct mscab_decompressor), sizeof(struct mschm_decompressor),
         offsetof(struct mschmd_header, sec1), offsetof(struct mscabd_file, folder));
  return 0;
}'''
    outs = []
    for inc in (os.path.dirname(ref), os.path.join(ROOT, "include")):
        with tempfile.TemporaryDirectory() as td:
            src = os.path.join(td, "t.c"); open(src, "w").write(prog)
            exe = os.path.join(td, "t")
            subprocess.check_call(["gcc", "-D_FILE_OFFSET_BITS=64", "-I", inc, src, "-o", exe])
            outs.append(subprocess.check_output([exe]).decode())
    assert outs[0] == outs[1]


def test_versions_and_selftest(built):
    L = api._setup()
    assert L.mspack_version(2) == 2 and L.mspack_version(4) == 2      # MSCABD, MSCHMD (system.c:16-51)
    assert L.mspack_version(0) == 1 and L.mspack_version(99) == -1
    assert L.mspack_sys_selftest_internal(8) == 0


def test_no_gpu_means_loud_failure_not_fallback(built):
    """On a machine without a GPU the batch call must FAIL (negative hip error), never decode."""
    import numpy as np
    if M.lib().mspack_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    units, out_bytes = M.make_units(M.KIND_LZX, [0], [16], [32768], window_bits=15)
    with pytest.raises(M.MspackHipError):
        M.decode_batch(units, np.zeros(64, dtype=np.uint8), out_bytes)


def test_no_gpu_job_fails_loudly_too(built):
    """The job entry points (include/mspack_hip.h) on a machine without a GPU: the batch fails on its thread, every wait and the end
    say so (negative hip error, mspack_hip_last_error set on the caller's thread), nothing is decoded -- and through the object API a
    cabinet's extract() answers MSPACK_ERR_DECRUNCH with the driver's "GPU batch decode failed" line, as the synchronous call does."""
    import zlib
    import numpy as np
    L = M.lib()
    if L.mspack_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    units, out_bytes = M.make_units(M.KIND_LZX, [0, 32], [16, 16], [32768, 32768], window_bits=15)
    arena = np.zeros(128, dtype=np.uint8); out = np.full(out_bytes + 64, 0x5A, dtype=np.uint8)
    res = np.zeros(2, dtype=M.RESULT_DTYPE)
    job = L.mspack_hip_decode_batch_begin(units.ctypes.data, 2, arena.ctypes.data, arena.size, out.ctypes.data, out.size, res.ctypes.data)
    assert job
    assert L.mspack_hip_job_wait_unit(job, 0) != 0 and L.mspack_hip_last_error()
    assert L.mspack_hip_job_wait_unit(job, 1) != 0
    assert L.mspack_hip_job_end(job) < 0
    assert (out == 0x5A).all()
    from libmspack_amd import api
    plain = M.gen_plaintext(3, 0, 2 * 32768)
    folders, files = [], []
    for i in range(2):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        folders.append((1, [b"CK" + co.compress(plain[i * 32768:(i + 1) * 32768].tobytes()) + co.flush()], [32768]))
        files.append((b"n%d.bin" % i, 32768, 0, i))
    with api.Cab(M.cab_write(folders, files), mem=True) as c:
        assert c.open_error == 0
        for i in (0, 1, 0):
            err, data = c.extract(i)
            assert err == 11 and len(data) == 0, (i, err, len(data))          # MSPACK_ERR_DECRUNCH
        assert any(b"GPU batch decode failed" in (m if isinstance(m, bytes) else str(m).encode()) for m in c.mem.messages), c.mem.messages


def test_open_of_missing_files_is_an_error_not_a_crash(built):
    """open() of a file that does not exist returns NULL with MSPACK_ERR_OPEN for every decompressor kind, also when
    malloc hands out dirty memory (MALLOC_PERTURB_: kwaj_open once freed two uninitialised pointers there)."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, sys
sys.path.insert(0, %r)
import libmspack_amd as M
L = M.lib()
for kind in ("cab", "chm", "szdd", "kwaj"):
    create = getattr(L, "mspack_create_%%s_decompressor" %% kind); destroy = getattr(L, "mspack_destroy_%%s_decompressor" %% kind)
    create.restype = C.c_void_p; create.argtypes = [C.c_void_p]; destroy.argtypes = [C.c_void_p]
    d = create(None)
    assert d
    vt = C.cast(d, C.POINTER(C.c_void_p))
    OPEN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_char_p)
    for _ in range(3):
        h = OPEN(vt[0])(d, b"/nonexistent/really/not/there.bin")
        assert not h
    n_err = {"cab": 7, "chm": 3, "szdd": 4, "kwaj": 4}[kind]          # index of last_error in the method table
    ERR = C.CFUNCTYPE(C.c_int, C.c_void_p)
    assert ERR(vt[n_err])(d) == 2, kind                                # MSPACK_ERR_OPEN
    destroy(d)
print("OPEN_OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MALLOC_PERTURB_="165")
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert p.returncode == 0 and b"OPEN_OK" in p.stdout, p.stdout.decode()[-2000:]
