import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import libmspack_amd as M
n, ub = 256, 65536
plain, comp, off, ln = M.corpus_lzx_units(0xBA5E11, 0, n, ub, 21)
units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2)
out, res = M.decode_batch(units, comp, out_bytes)
raw = res.view(np.uint32).reshape(n, 6).astype(np.int64)
tot = raw[:, 3].mean()
print("total(ticks/64)", tot, " rounds/unit:", (raw[:, 5] >> 16).mean(), " with vectorised copies:", (raw[:, 5] & 0xFFFF).mean())
for name, c in [("vector decode", 1), ("chain walk", 2), ("scan+literals", 4), ("checks+copies", 0)]:
    v = raw[:, c].mean(); print("  %-16s %10.0f  %5.1f%%" % (name, v, 100 * v / tot))
