# scratch: randomized differential sweep of the MSZIP kernel against the CPU oracle (GPU box, via gpurun):
# random deflate levels / strategies, block sizes (incl. short blocks with cross-block history), plaintext
# families, shorter requests, damage, repair mode with random feeder chunk sizes.
import sys, zlib, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import libmspack_amd as M
from helpers import oracle_mszip
from test_gpu_mszip import folder
from test_gpu_fuzz import mutations
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
streams, lens, flags, chunks = [], [], [], []
for c in range(n_cfg):
    n = int(rng.integers(1, 250000))
    data = M.gen_plaintext(5000 * seed + c, int(rng.integers(0, 6)), n).tobytes()
    level = int(rng.integers(0, 10))
    strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    bs = 32768 if rng.random() < .6 else int(rng.integers(1, 32769))
    hist = bool(rng.random() < .6) and bs == 32768      # history is only well defined after full blocks
    s = folder(data, level, strat, history=hist, bs=bs)
    for k, m in enumerate([s, s] + mutations(s, rng, 8)):
        rep = rng.random() < .3
        streams.append(m); flags.append(M.UF_MSZIP_REPAIR if rep else 0)
        chunks.append(int(rng.choice([0, 2, 64, 512, 1000, 4096])) if rep else 0)
        lens.append(n if k != 1 else int(rng.integers(0, n + 1)))
offs, pos = [], 0
for s in streams:
    pos = (pos + 15) & ~15
    offs.append(pos); pos += len(s)
arena = np.zeros(pos + 64, dtype=np.uint8)
for s, o in zip(streams, offs):
    arena[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
units, out_bytes = M.make_units(M.KIND_MSZIP, offs, [len(s) for s in streams], lens, flags=flags, out_slack=32768)
units["in_chunk"] = chunks
out, res = M.decode_batch(units, arena, out_bytes)
bad = 0
for i, st in enumerate(streams):
    rp = 0 if not flags[i] else (chunks[i] if chunks[i] else 1)
    e, o, r, _ = oracle_mszip(st, lens[i], rp)
    got = out[units["out_off"][i]:units["out_off"][i] + r.out_len].tobytes()
    if res["err"][i] != e or res["out_len"][i] != r.out_len or got != o[:r.out_len]:
        bad += 1
        print("MISMATCH unit", i, lens[i], flags[i], chunks[i], "gpu", res[i], "oracle", e, r.out_len)
print("seed", seed, "units", len(streams), "gpu errors", int((res["err"] != 0).sum()), "mismatches", bad)
