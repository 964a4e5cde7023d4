import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import libmspack_amd as M
n, ub = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 65536
plain, comp, off, ln = M.corpus_lzx_units(0xC0FFEE, M.TEXT_MIX, n, ub, 21)
units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2)
out, res = M.decode_batch(units, comp, out_bytes)
bad = 0
for i in range(n):
    o = out[i * ub:(i + 1) * ub]; p = plain[i * ub:(i + 1) * ub]
    if res["err"][i] != 0 or not np.array_equal(o, p):
        d = np.nonzero(o != p)[0]
        print("unit", i, "res", res[i], "in_len", ln[i] + 4, "first diff", d[:5] if len(d) else None, "ndiff", len(d))
        if len(d):
            k = int(d[0]); print("   got", bytes(o[k-8:k+16]), "\n   exp", bytes(p[k-8:k+16]))
        bad += 1
        if bad > 6: break
print("bad", bad, "of", n)
