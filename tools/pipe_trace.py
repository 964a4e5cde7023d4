"""scratch (analysis build -DLZX_PIPE_TRACE): per-ticket timeline of one mspack_lzx_pipe launch.
MSPACK_HIP_SO=build/variants/libmspack_hip_trace.so python tools/pipe_trace.py [units] [out.npy]"""
import sys, os, ctypes
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import libmspack_amd as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ub = 65536
plain, comp, off, ln, tab = M.corpus_lzx_units(0xBA5E11, 0, n, ub, 21, frame_tables=True)
units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2, frame_tabs=tab)
L = M.lib()
ph = np.zeros(32, dtype=np.uint64)
if hasattr(L, "mspack_hip_debug_pipe_phases"):
    L.mspack_hip_debug_pipe_phases.argtypes = [ctypes.c_void_p]
for it in range(3):
    if it == 2 and hasattr(L, "mspack_hip_debug_pipe_phases"):
        L.mspack_hip_debug_pipe_phases(ph.ctypes.data)          # clear: the phases of the last launch only
    out, res = M.decode_batch(units, comp, out_bytes)
if hasattr(L, "mspack_hip_debug_pipe_phases"):
    L.mspack_hip_debug_pipe_phases(ph.ctypes.data)
    names = ["P wait prev header", "P header decode", "P record + publish", "P table builds", "P parse_emit total", "P final publish",
             "  emit: staging", "  emit: sync + count rounds", "  emit: last walk (values, literals, records)"]
    print("parse tasks, us per frame (%d frames):" % (2 * n))
    for k, nm in enumerate(names):
        print("  %-46s %8.1f" % (nm, ph[k] / 100.0 / (2 * n)))
    if ph[15]:
        print("  per pass: %.1f steps of the count walks in %.2f rounds, %.1f steps of the last walk; %.2f passes per frame" %
              (ph[12] / ph[15], ph[13] / ph[15], ph[14] / ph[15], ph[15] / (2.0 * n)))
        print("            of the count walks' steps %.1f are the walks behind the second (%.2f of them per pass, %.2f lanes walking in each)" % (ph[9] / ph[15], ph[13] / ph[15] - 2.0, ph[10] / max(1.0, float(ph[13]) - 2.0 * float(ph[15]))))
    print("resolve half of the tasks, us per UNIT (two frames): front (load, R0-R2, checks) %.1f  push %.1f  resolve %.1f" %
          (ph[16 + 9] / 100.0 / n, ph[16 + 10] / 100.0 / n, ph[16 + 11] / 100.0 / n))
assert (res["err"] == 0).all() and np.array_equal(out[:n * ub], plain)
T = min(4 * n, 1 << 16)
a = np.zeros(4 * T, dtype=np.uint64)
L.mspack_hip_debug_pipe_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert L.mspack_hip_debug_pipe_trace(a.ctypes.data, a.size) == 0
a = a.reshape(T, 4)
if len(sys.argv) > 2:
    np.save(sys.argv[2], a)
v = a[a[:, 1] != 0]
t0 = v[:, 0].min()
st = (v[:, 0] - t0) / 100.0; en = (v[:, 1] - t0) / 100.0            # us (100 MHz)
kind = (v[:, 2] & 1).astype(int); fr = ((v[:, 2] & 0xFFFFFFFF) >> 1).astype(int)
wait = (v[:, 3] & ((1 << 40) - 1)) / 100.0; blk = (v[:, 3] >> 40).astype(int)
print("adopted %.3f; launch span %.0f us; %d tasks traced" % (((res["flags"] & 32) != 0).mean(), en.max(), len(v)))
for k, f, name in ((0, 0, "P frame 0"), (0, 1, "P frame 1"), (1, 0, "R frame 0"), (1, 1, "R frame 1")):
    m = (kind == k) & ((fr == f) if f is not None else True)
    if m.any():
        d = en[m] - st[m]
        print("%-10s n %5d  start %7.0f..%7.0f  end %7.0f..%7.0f  dur mean %6.0f min %6.0f max %6.0f us  waited mean %6.0f max %6.0f" %
              (name, m.sum(), st[m].min(), st[m].max(), en[m].min(), en[m].max(), d.mean(), d.min(), d.max(), wait[m].mean(), wait[m].max()))
busy = np.zeros(blk.max() + 1)
np.add.at(busy, blk, en - st - wait)
print("waves %d; busy (task time minus waits) per wave: mean %.0f us = %.2f of the span; sum of task time %.0f wave-us" %
      (len(busy), busy.mean(), busy.mean() / en.max(), (en - st).sum()))
h, e = np.histogram(en, bins=12)
print("task ends per %.0f us: %s" % (e[1] - e[0], h.tolist()))
