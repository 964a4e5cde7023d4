"""GPU parity for Quantum: streams from our encoder (validated against the reference in the dev
container) vs. the CPU oracle, bit-exact; long enough to hit model rescales and the every-50th
re-sort of each model (qtmd.c:125-166); small windows exercise matches that wrap the window."""
import numpy as np
import pytest

import libmspack_amd as M
from helpers import oracle_qtm

pytestmark = pytest.mark.gpu


def run(streams, out_lens, wbs):
    offs, pos = [], 0
    for s in streams:
        pos = (pos + 15) & ~15
        offs.append(pos); pos += len(s)
    arena = np.zeros(pos + 64, dtype=np.uint8)
    for s, o in zip(streams, offs):
        arena[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
    units, out_bytes = M.make_units(M.KIND_QUANTUM, offs, [len(s) for s in streams], out_lens, window_bits=wbs)
    out, res = M.decode_batch(units, arena, out_bytes)
    return units, out, res


def test_qtm_kinds_and_windows(built):
    streams, lens, wbs, plains = [], [], [], []
    for kind in range(6):
        for wb, n in [(21, 300000), (18, 100000), (10, 70000), (15, 33000), (12, 2000)]:
            d = M.gen_plaintext(50 + kind, kind, n)
            s, _fs = M.qtm_encode(d, wb)
            streams.append(s); lens.append(n); wbs.append(wb); plains.append(d)
    units, out, res = run(streams, lens, wbs)
    for i, s in enumerate(streams):
        e, o, r = oracle_qtm(s, lens[i], wbs[i])
        assert res["err"][i] == e == 0, (i, res[i], e)
        assert res["out_len"][i] == r.out_len
        got = out[units["out_off"][i]:units["out_off"][i] + lens[i]]
        assert got.tobytes() == o, "unit %d (kind %d wb %d) differs at %d" % (
            i, i // 5, wbs[i], int(np.nonzero(got != np.frombuffer(o, dtype=np.uint8))[0][0]))
        assert np.array_equal(got, plains[i])


def test_qtm_partial_truncated_corrupt(built):
    d = M.gen_plaintext(9, M.TEXT_MIX, 120000)
    s, _ = M.qtm_encode(d, 16)
    rng = np.random.default_rng(5)
    streams, lens = [], []
    for want in (1, 1000, 32768, 32769, 65536, 65537, 119999):
        streams.append(s); lens.append(want)
    for cut in (0, 1, 2, 3, 100, len(s) // 2, len(s) - 2, len(s) - 1):
        streams.append(s[:cut]); lens.append(d.size)
    for _ in range(30):
        b = bytearray(s); k = int(rng.integers(0, len(b))); b[k] ^= 1 << int(rng.integers(0, 8))
        streams.append(bytes(b)); lens.append(d.size)
    units, out, res = run(streams, lens, [16] * len(streams))
    for i, st in enumerate(streams):
        e, o, r = oracle_qtm(st, lens[i], 16)
        assert res["err"][i] == e, (i, res[i], e)
        assert res["out_len"][i] == r.out_len, (i, res[i], r.out_len)
        if e == 0:
            got = out[units["out_off"][i]:units["out_off"][i] + r.out_len].tobytes()
            assert got == o[:r.out_len], i


def test_qtm_batch_512_folders(built):
    """BASELINE config 4 as SURVEY 8(d) specifies it: 512 independent folders of 32 blocks (1 MiB) each,
    window 21 (comp_type 0x1572)."""
    from concurrent.futures import ThreadPoolExecutor
    n, ub = 512, 32 * 32768
    plain = M.gen_plaintext(123, M.TEXT_MIX, n * ub)
    with ThreadPoolExecutor(max_workers=16) as ex:
        streams = list(ex.map(lambda i: M.qtm_encode(plain[i * ub:(i + 1) * ub], 21)[0], range(n)))
    units, out, res = run(streams, [ub] * n, [21] * n)
    assert (res["err"] == 0).all() and (res["out_len"] == ub).all()
    for i in range(n):
        o = units["out_off"][i]
        assert np.array_equal(out[o:o + ub], plain[i * ub:(i + 1) * ub]), i
