# scratch (GPU box): host-buffer entry points on the headline batch; MSPACK_HIP_NSTREAMS / _TRACE from the environment
import sys, time, ctypes as C, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import libmspack_amd as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ub = 65536
plain, comp, off, ln = M.corpus_lzx_units(0xBA5E11, 0, n, ub, 21)
units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2)
L = M.lib()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), out_bytes + 64) == 0
res = np.zeros(n, dtype=M.RESULT_DTYPE)
u = np.ascontiguousarray(units)
h_out = np.zeros(out_bytes + 64, dtype=np.uint8)
for it in range(reps):
    t0 = time.perf_counter()
    rc = L.mspack_hip_decode_batch_to_device(u.ctypes.data, n, comp.ctypes.data, comp.size, p.value, out_bytes + 64, res.ctypes.data)
    dt = time.perf_counter() - t0
    assert rc == 0 and (res["err"] == 0).all()
    print("to_device: %.2f ms  %.1f GB/s" % (dt * 1e3, n * ub / dt / 1e9), flush=True)
for it in range(reps):
    t0 = time.perf_counter()
    rc = L.mspack_hip_decode_batch(u.ctypes.data, n, comp.ctypes.data, comp.size, h_out.ctypes.data, out_bytes + 64, res.ctypes.data)
    dt = time.perf_counter() - t0
    assert rc == 0 and (res["err"] == 0).all()
    print("to_host:   %.2f ms  %.1f GB/s%s" % (dt * 1e3, n * ub / dt / 1e9, "" if np.array_equal(h_out[:n * ub], plain) else "  MISMATCH"), flush=True)
