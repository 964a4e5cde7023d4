# scratch (GPU box): host-buffer entry points on the headline batch (frame tables on), one process per MSPACK_HIP_NCHUNKS value
#   python tools/exp_hostpath.py [n_units] [reps] [chunks,chunks,...]        (MSPACK_HIP_TRACE=1 for the library's own split)
import os, subprocess, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if "EXP_HOSTPATH_WORKER" not in os.environ:
    for nc in (sys.argv[3] if len(sys.argv) > 3 else "1,2,4,8").split(","):
        print("== MSPACK_HIP_NCHUNKS=%s" % nc, flush=True)
        subprocess.call([sys.executable, __file__] + sys.argv[1:3], env=dict(os.environ, EXP_HOSTPATH_WORKER="1", MSPACK_HIP_NCHUNKS=nc))
    sys.exit(0)
import numpy as np
if os.environ.get("EXP_TORCH"):              # a process that has streams of its own (as bench.py has)
    import torch
    _t = torch.zeros(1 << 20, device="cuda"); _s = [torch.cuda.Stream() for _ in range(int(os.environ["EXP_TORCH"]))]; torch.cuda.synchronize()
sys.path.insert(0, ROOT + '/tests'); sys.path.insert(0, ROOT)
import libmspack_amd as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ub = 65536
plain, comp, off, ln, tab = M.corpus_lzx_units(0xBA5E11, 0, n, ub, 21, frame_tables=True)
units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2, frame_tabs=tab)
L = M.lib()
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), out_bytes + 64) == 0
res = np.zeros(n, dtype=M.RESULT_DTYPE)
u = np.ascontiguousarray(units)
h_out = np.zeros(out_bytes + 64, dtype=np.uint8)
ts = []
for it in range(reps + 1):
    t0 = time.perf_counter()
    rc = L.mspack_hip_decode_batch_to_device(u.ctypes.data, n, comp.ctypes.data, comp.size, p.value, out_bytes + 64, res.ctypes.data)
    ts.append(time.perf_counter() - t0)
    assert rc == 0 and (res["err"] == 0).all()
print("to_device: " + " ".join("%.2f" % (t * 1e3) for t in ts) + " ms   best %.1f GB/s" % (n * ub / min(ts) / 1e9), flush=True)
ts = []
for it in range(reps + 1):
    h_out[:n * ub:4096] = 0
    t0 = time.perf_counter()
    rc = L.mspack_hip_decode_batch(u.ctypes.data, n, comp.ctypes.data, comp.size, h_out.ctypes.data, out_bytes + 64, res.ctypes.data)
    ts.append(time.perf_counter() - t0)
    assert rc == 0 and (res["err"] == 0).all()
    assert np.array_equal(h_out[:n * ub], plain), "MISMATCH"
print("to_host:   " + " ".join("%.2f" % (t * 1e3) for t in ts) + " ms   best %.1f GB/s (bit-exact)" % (n * ub / min(ts) / 1e9), flush=True)
