import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import libmspack_amd as M
n, ub = 256, 65536
plain, comp, off, ln = M.corpus_lzx_units(0xBA5E11, 0, n, ub, 21)
units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2)
out, res = M.decode_batch(units, comp, out_bytes)
tot = res["in_used"].mean()
print("total(ticks/64)", tot)
for name, f in [("vector decode", "flags"), ("chain walk", "out_len"), ("scan+literals", "good_len"), ("R pass", "err"), ("checks+copies", "reserved")]:
    v = res[f].astype(np.int64).mean(); print("  %-16s %10.0f  %5.1f%%" % (name, v, 100 * v / tot))
