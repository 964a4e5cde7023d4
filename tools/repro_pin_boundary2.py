"""scratch (GPU box): round 5 -- the sequence around the one abort of test_copies_are_cut_at_pin_boundaries in a whole-suite run:
the headline batch through both host entry points, mspack_hip_release(), a small batch, then the partial-lock batches; N times."""
import ctypes as C, os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import libmspack_amd as M
from test_gpu_hostpath import DevBuf
n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 6
L = M.lib()
L.mspack_hip_pin.argtypes = [C.c_void_p, C.c_size_t]
L.mspack_hip_unpin.argtypes = [C.c_void_p]
N, ub = 4096, 65536
plainH, compH, offH, lnH = M.corpus_lzx_units(0xBA5E11, 0, N, ub, 21)
unitsH, obH = M.make_units(M.KIND_LZX, offH, lnH + 4, np.full(N, ub), window_bits=21, reset_frames=2)
n = 256
plain, comp, off, ln = M.corpus_lzx_units(0x9191, 0, n, ub, 21)
units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2)
arena = np.zeros(comp.size + 8192, dtype=np.uint8)
for rep in range(n_rep):
    out, res = M.decode_batch(unitsH, compH, obH)
    assert (res["err"] == 0).all() and np.array_equal(out[:N * ub], plainH)
    d_out = DevBuf(obH + 64)
    res2 = np.zeros(N, dtype=M.RESULT_DTYPE); uH = np.ascontiguousarray(unitsH)
    rc = L.mspack_hip_decode_batch_to_device(uH.ctypes.data, N, compH.ctypes.data, compH.size, d_out.ptr, obH + 64, res2.ctypes.data)
    assert rc == 0
    d_out.free()
    L.mspack_hip_release()
    out, res = M.decode_batch(unitsH[:64], compH, obH)
    del out
    print("rep %d: headline part done" % rep, flush=True, file=sys.stderr)
    for shift, lo_frac, hi_frac in ((0, 0.0, 0.5), (100, 0.25, 0.75), (4000, 0.5, 1.0)):
        a = arena[shift:shift + comp.size]
        a[:] = comp
        out = np.zeros(out_bytes + 4096 + 64, dtype=np.uint8)[shift % 64:]
        p_in = a.ctypes.data + int(comp.size * lo_frac)
        p_out = out.ctypes.data + int(out_bytes * lo_frac)
        r_in = L.mspack_hip_pin(p_in, int(comp.size * (hi_frac - lo_frac)))
        r_out = L.mspack_hip_pin(p_out, int(out_bytes * (hi_frac - lo_frac)))
        resx = np.zeros(n, dtype=M.RESULT_DTYPE)
        u = np.ascontiguousarray(units)
        print("rep %d shift %d: pins %d %d ..." % (rep, shift, r_in, r_out), end=" ", flush=True, file=sys.stderr)
        rc = L.mspack_hip_decode_batch(u.ctypes.data, n, a.ctypes.data, a.size, out.ctypes.data, out_bytes + 64, resx.ctypes.data)
        ok = rc == 0 and (resx["err"] == 0).all() and np.array_equal(out[:n * ub], plain)
        print("rc %d %s %s" % (rc, "ok" if ok else "WRONG", L.mspack_hip_last_error().decode()), flush=True, file=sys.stderr)
        L.mspack_hip_unpin(p_in); L.mspack_hip_unpin(p_out)
print("done", file=sys.stderr)
