# scratch: randomized differential sweep of the MSZIP block-parallel path (block parse waves with the deflate lane parser,
# literals stored by the parse waves, match records committed by the folder's wave) against the CPU oracle: random levels,
# strategies, block sizes, histories, plaintext families, shorter requests, damage -- every folder with its block table,
# a third of them with a table that is wrong.   python tools/sweep_mszip_blocks.py <seed> [configs]
import sys, zlib, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import libmspack_amd as M
from helpers import oracle_mszip
from test_gpu_mszip_blocks import folder_blocks, run, ADOPTED
from test_gpu_fuzz import mutations
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
streams, lens, tabs, clean = [], [], [], []
for c in range(n_cfg):
    n = int(rng.integers(1, 260000))
    data = M.gen_plaintext(7000 * seed + c, int(rng.integers(0, 6)), n).tobytes()
    level = int(rng.integers(0, 10))
    strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
    bs = 32768 if rng.random() < .7 else int(rng.integers(1, 32769))
    hist = bool(rng.random() < .6) and bs == 32768
    se = int(rng.choice([0, 0, 0, 2, 3]))
    s, offs = folder_blocks(data, level, strat, history=hist, bs=bs, stored_every=se)
    offs = np.array(offs, dtype=np.int64)
    def table():
        r = rng.random()
        if r < .66: return offs
        if r < .8: return offs + int(rng.integers(-3, 8))
        if r < .9: return rng.integers(0, max(1, len(s)), offs.size)
        return np.sort(rng.integers(0, max(1, len(s)), offs.size))
    streams.append(s); lens.append(n); tabs.append(offs); clean.append(True)
    streams.append(s); lens.append(int(rng.integers(0, n + 1))); tabs.append(table()); clean.append(False)
    for m in mutations(s, rng, 6):
        streams.append(m); lens.append(n); tabs.append(table()); clean.append(False)
# a table must fit the unit's blocks: ceil(out_len / 32768) entries (shorter tables are padded with their last entry)
fixed = []
for t, n in zip(tabs, lens):
    nb = max(1, (n + 32767) // 32768)
    t = np.asarray(t, dtype=np.int64)
    if t.size < nb: t = np.concatenate([t, np.full(nb - t.size, t[-1] if t.size else 0)])
    fixed.append(np.clip(t[:nb], 0, 2**31 - 1))
units, out, res = run(streams, lens, fixed)
bad = 0
for i, st in enumerate(streams):
    e, o, r, _ = oracle_mszip(st, lens[i])
    got = out[units["out_off"][i]:units["out_off"][i] + r.out_len].tobytes()
    same = got == o[:r.out_len] or e != 0                    # damaged streams may read window bytes the reference never wrote
    if res["err"][i] != e or res["out_len"][i] != r.out_len or not same:
        bad += 1
        print("MISMATCH unit", i, lens[i], "gpu", res[i], "oracle", e, r.out_len)
print("seed", seed, "units", len(streams), "adopted", int(((res["flags"] & ADOPTED) != 0).sum()), "errors in oracle",
      sum(1 for i, st in enumerate(streams) if oracle_mszip(st, lens[i])[0] != 0), "mismatches", bad)
