import sys, os, ctypes as C, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import libmspack_amd as M, helpers
n, ub = 4096, 65536
plain, comp, off, ln = M.corpus_lzx_units(0xBA5E11, 0, n, ub, 21)
R = helpers.ref()
off64 = np.ascontiguousarray(off, dtype=np.uint64); ilen = np.ascontiguousarray(ln + 4, dtype=np.uint32); olen = np.full(n, ub, dtype=np.uint32)
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try: print("cgroup cpu.max:", open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e: print("cgroup", e)
for th in (1, 4, 8, 16, 32, 64, 128, 256):
    b = C.c_ulonglong(0); e = C.c_int(0)
    reps = max(1, th // 4)
    t = R.refh_bench(0, comp.ctypes.data, off64.ctypes.data, ilen.ctypes.data, olen.ctypes.data, n, 21, 2, th, reps, C.byref(b), C.byref(e))
    print("threads %3d: %8.1f MB/s (%.2fs, errors %d)" % (th, b.value / t / 1e6, t, e.value))
