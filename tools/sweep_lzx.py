# scratch: randomized differential sweep of the LZX kernel against the CPU oracle (GPU box, via gpurun):
# random windows, reset intervals, block modes and sizes, plaintext families, unit cuts, damage.
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import libmspack_amd as M
from helpers import oracle_lzx
from test_gpu_lzx import run_units
from test_gpu_fuzz import mutations
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
streams, params = [], []
for c in range(n_cfg):
    wb = int(rng.integers(15, 22)); reset = int(rng.choice([0, 0, 1, 2, 3, 4]))
    n = int(rng.integers(1, 200000))
    kind = int(rng.integers(0, 6))
    kw = {}
    m = int(rng.integers(0, 5))
    if m: kw["mode"] = m
    if m in (0, 4): kw["block_size"] = int(rng.integers(1, 70000))
    if rng.random() < .2: kw["intel_filesize"] = int(rng.integers(1, 400000))
    if rng.random() < .2: kw["repeats"] = 0
    if rng.random() < .2: kw["lazy"] = 0
    data = M.gen_plaintext(1000 * seed + c, kind, n)
    try:
        comp = M.lzx_encode(data, wb, reset, M.lzx_opts(**kw))[0].tobytes()
    except M.MspackHipError:
        continue                                            # the corpus encoder's output estimate (tiny blocks of noise)
    tail = b"\0" * 4 if reset else b""
    streams.append(comp + tail); params.append((n, wb, reset, 0))
    for mu in mutations(comp, rng, 8):
        streams.append(mu + tail); params.append((n, wb, reset, 0))
    cut = int(rng.integers(0, n + 1))                       # a shorter request (lzxd_decompress(out_bytes) semantics)
    streams.append(comp + tail); params.append((cut, wb, reset, 0))
units, out, res = run_units(streams, params)
bad = 0
for i, (s, p) in enumerate(zip(streams, params)):
    e, o, r = oracle_lzx(s, p[0], p[1], p[2], length=p[0], e8_base=p[3])
    got = out[units["out_off"][i]:units["out_off"][i] + r.out_len].tobytes()
    if res["err"][i] != e or res["out_len"][i] != r.out_len or res["flags"][i] != r.flags or got != o[:r.out_len]:
        bad += 1
        print("MISMATCH unit", i, p, "gpu", res[i], "oracle", e, r.out_len, r.flags)
print("seed", seed, "units", len(streams), "errors in oracle", int((res["err"] != 0).sum()), "mismatches", bad)
