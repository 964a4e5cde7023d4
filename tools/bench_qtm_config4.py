"""scratch (GPU box): BASELINE config 4 -- 512 Quantum folders of 32 blocks, window 2^21 -- kernel time of the library named by
MSPACK_HIP_SO (default: the shipped one), every byte verified.  python tools/bench_qtm_config4.py [folders] [frames] [marks per folder]"""
import os, sys, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import torch
import libmspack_amd as M
import bench as B
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
fr = int(sys.argv[2]) if len(sys.argv) > 2 else 32
mk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
r = B.secondary_qtm(M, torch, torch.device("cuda", 0), n=n, frames=fr, cpu=False, marks=mk)
print(json.dumps({k: r[k] for k in ("kernel_ms", "value", "bit_exact")}), os.environ.get("MSPACK_HIP_SO", "shipped"), "marks per folder: %d" % mk)

L = M.lib()
if hasattr(L, "mspack_hip_debug_qtm_timers"):
    import ctypes
    t = np.zeros(8, dtype=np.uint64)
    L.mspack_hip_debug_qtm_timers.argtypes = [ctypes.c_void_p]
    L.mspack_hip_debug_qtm_timers(t.ctypes.data)
    ns, tot, P = float(t[4]), float(t[5]), float(t[6])
    print("block 0's wave: %d symbols, %d output bytes, %.0f cycles (s_memtime) in all = %.0f per symbol, %.0f per byte;" % (ns, P, tot, tot / ns, tot / P))
    print("  per symbol: interval %.0f, model update %.0f, renormalisation + bits %.0f, outside GET_SYMBOL %.0f" %
          (t[0] / ns, t[1] / ns, t[2] / ns, (tot - float(t[0] + t[1] + t[2])) / ns))
