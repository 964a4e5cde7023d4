# scratch: event counters of the LZX speculative path (build with -DLZX_EXP_CNT, run with MSPACK_HIP_SO=...)
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import libmspack_amd as M
n, ub = 256, 65536
plain, comp, off, ln = M.corpus_lzx_units(0xBA5E11, 0, n, ub, 21)
units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2)
out, res = M.decode_batch(units, comp, out_bytes)
raw = res.view(np.uint32).reshape(n, 6).astype(np.int64)
names = {0: "rounds in which some lane (on the chain or not) had a long main code", 1: "tokens the scalar decoder took", 2: "commits", 3: "on-chain tokens with a main code beyond the direct table", 4: "commits with the general R0-R2 scan", 5: "rounds"}
for k in range(6): print(names[k], raw[:, k].mean())
