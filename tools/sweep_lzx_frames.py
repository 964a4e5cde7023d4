# scratch: randomized differential sweep of the frame-parallel LZX path (header / parse waves, lane parser, literal
# runs, adoption by the unit wave) against the CPU oracle: random windows, reset intervals, block modes and sizes,
# plaintext families, damage -- every unit with its frame table, a third of them with a table that is wrong.
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import libmspack_amd as M
from helpers import oracle_lzx
from test_gpu_lzx_frames import run, ADOPTED
from test_gpu_fuzz import mutations
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
streams, params, tabs = [], [], []
for c in range(n_cfg):
    wb = int(rng.integers(15, 22)); reset = int(rng.choice([0, 0, 1, 2, 3, 4]))
    n = int(rng.integers(1, 300000))
    kind = int(rng.integers(0, 6))
    kw = {}
    m = int(rng.integers(0, 5))
    if m: kw["mode"] = m
    if m in (0, 4) and rng.random() < .5: kw["block_size"] = int(rng.integers(1, 70000))
    if rng.random() < .2: kw["intel_filesize"] = int(rng.integers(1, 400000))
    if rng.random() < .2: kw["repeats"] = 0
    if rng.random() < .2: kw["lazy"] = 0
    data = M.gen_plaintext(1000 * seed + c, kind, n)
    try:
        comp, fo = M.lzx_encode(data, wb, reset, M.lzx_opts(**kw))
    except M.MspackHipError:
        continue
    comp = comp.tobytes(); fo = fo.astype(np.int64)[:-1]
    tail = b"\0" * 4 if reset else b""
    def table():
        r = rng.random()
        if r < .66 or fo.size == 0: return fo
        if r < .8: return fo + int(rng.integers(-8, 64)) * 2
        if r < .9: return rng.integers(0, max(1, len(comp)), fo.size)
        return np.sort(rng.integers(0, max(1, len(comp)), fo.size))
    streams.append(comp + tail); params.append((n, wb, reset, 0)); tabs.append(fo)
    for mu in mutations(comp, rng, 6):
        streams.append(mu + tail); params.append((n, wb, reset, 0)); tabs.append(table())
    cut = int(rng.integers(0, n + 1))                       # a shorter request (lzxd_decompress(out_bytes) semantics)
    streams.append(comp + tail); params.append((cut, wb, reset, 0)); tabs.append(table())
units, out, res = run(streams, params, tabs)
bad = 0
for i, (s, p) in enumerate(zip(streams, params)):
    e, o, r = oracle_lzx(s, p[0], p[1], p[2], length=p[0], e8_base=p[3])
    got = out[units["out_off"][i]:units["out_off"][i] + r.out_len].tobytes()
    same_bytes = got == o[:r.out_len] or e != 0              # damaged streams may read window bytes the reference never wrote
    if res["err"][i] != e or res["out_len"][i] != r.out_len or (int(res["flags"][i]) & ~ADOPTED) != r.flags or \
       res["in_next"][i] != r.in_next or not same_bytes:
        bad += 1
        print("MISMATCH unit", i, p, "gpu", res[i], "oracle", e, r.out_len, r.flags, r.in_next)
print("seed", seed, "units", len(streams), "adopted", int(((res["flags"] & ADOPTED) != 0).sum()),
      "errors in oracle", int((res["err"] != 0).sum()), "mismatches", bad)
