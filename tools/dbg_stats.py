import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import libmspack_amd as M
n, ub = 256, 65536
plain, comp, off, ln = M.corpus_lzx_units(0xBA5E11, 0, n, ub, 21)
units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2)
out, res = M.decode_batch(units, comp, out_bytes)
print("err", set(res["err"]), "rounds/unit", res["in_used"].mean(), "unknown/unit", res["reserved"].mean())
