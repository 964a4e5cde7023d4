"""GPU parity for LZX DELTA: streams of our DELTA encoder (validated against the real
lzxd in tests/test_oracle_vs_ref.py) through the C ABI as MSPACK_HIP_KIND_LZX_DELTA units -- reference
data below the output, per-frame chunk sizes, extended match lengths, windows 2^17..2^25 -- against the
oracle: error code, flags, byte count and every byte; plus damaged copies."""
import numpy as np
import pytest

import libmspack_amd as M
from helpers import oracle_lzxd
from test_oracle_vs_ref import DELTA_CASES, delta_case, far_offset_case
from test_gpu_fuzz import mutations

pytestmark = pytest.mark.gpu


def run_delta(streams, params, refs):
    """params: (out_len, window_bits)"""
    offs, pos = [], 0
    for s in streams:
        pos = (pos + 15) & ~15
        offs.append(pos); pos += len(s)
    arena = np.zeros(pos + 64, dtype=np.uint8)
    for s, o in zip(streams, offs):
        arena[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
    units, out_bytes = M.make_units(M.KIND_LZX_DELTA, offs, [len(s) for s in streams], [p[0] for p in params],
                                    window_bits=[p[1] for p in params], ref_lens=[len(r) for r in refs])
    out, res = M.decode_batch(units, arena, out_bytes, refs=refs)
    return units, out, res


def test_lzx_delta_streams_and_damage(built):
    rng = np.random.default_rng(99)
    streams, params, refs, plains = [], [], [], []
    for (n, wb, refn, kw) in DELTA_CASES:
        data, ref, comp = delta_case(n, wb, refn, kw)
        variants = [comp + b"\0" * 8] + (mutations(comp, rng, 60) if n <= 400000 else [])
        for v in variants:
            streams.append(v); params.append((n, wb)); refs.append(ref); plains.append(data if v is variants[0] else None)
    units, out, res = run_delta(streams, params, refs)
    bad = 0
    for i, (s, p, r) in enumerate(zip(streams, params, refs)):
        e, o, rr = oracle_lzxd(s, p[0], p[1], r)
        assert res["err"][i] == e, (i, res[i], e)
        assert res["out_len"][i] == rr.out_len, (i, res[i], rr.out_len)
        assert res["flags"][i] == rr.flags, (i, res[i], rr.flags)
        got = out[units["out_off"][i]:units["out_off"][i] + rr.out_len].tobytes()
        assert got == o[:rr.out_len], "unit %d differs at byte %d" % (i, next(k for k in range(len(got)) if got[k] != o[k]))
        if plains[i] is not None:
            assert e == 0 and got == plains[i].tobytes()
        bad += e != 0
    assert bad > 50


def test_lzx_delta_window_limits(built):
    """windows lzxd_init refuses for DELTA streams (lzxd.c:288-293) answer MSPACK_ERR_ARGS"""
    data = M.gen_plaintext(1, 0, 40000)
    comp = M.lzxd_encode(data, 17).tobytes()
    units, out, res = run_delta([comp] * 3, [(40000, 16), (40000, 26), (40000, 17)], [b""] * 3)
    assert list(res["err"][:2]) == [1, 1] and res["err"][2] == 0


def test_lzx_delta_offsets_beyond_the_match_list_field(built):
    """A 2^25 window with matches 9.5 MB back: offsets above 2^23 do not fit the 23-bit offset field of a
    queued match (spec_queue.hpp) and must take the one-by-one copy path -- every byte against the
    plaintext and the oracle.  (The same stream decodes to the plaintext in the real lzxd:
    test_oracle_vs_ref.py::test_lzx_delta_far_offsets.)"""
    data, comp = far_offset_case()
    units, out, res = run_delta([comp + b"\0" * 8], [(data.size, 25)], [b""])
    e, o, rr = oracle_lzxd(comp + b"\0" * 8, data.size, 25)
    assert e == 0 and res["err"][0] == 0 and res["out_len"][0] == rr.out_len == data.size
    got = out[units["out_off"][0]:units["out_off"][0] + data.size]
    assert np.array_equal(got, data) and got.tobytes() == o
