"""GPU parity for Quantum: streams from our encoder (validated against the reference in the dev
container) vs. the CPU oracle, bit-exact; long enough to hit model rescales and the every-50th
re-sort of each model (qtmd.c:125-166); small windows exercise matches that wrap the window."""
import numpy as np
import pytest

import libmspack_amd as M
from helpers import oracle_qtm, oracle_qtm_marks

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


def test_qtm_marks_vs_oracle(built):
    """MSPACK_HIP_UF_QTM_MARKS (round 6): what a request ending at each marked position holds back for the next call (qtmd.c:268-276),
    0xFFFFFFFF where the reference cannot end a request (inside a match that crosses the window's end, qtmd.c:358-374), 0 where the
    stream never got to -- against ONE oracle decode with the same marks (the oracle's marks are pinned to the real qtmd's
    o_end - o_ptr in tests/test_oracle_vs_ref.py): every window from 10 to 21, marks at random places and around every window
    end, clean, damaged and cut streams, one unit whose marks outnumber its 32 KiB of slack."""
    rng = np.random.default_rng(77)
    streams, lens, wbs, marks = [], [], [], []
    for wb in range(10, 22):
        n = int(rng.integers(20000, 90000))
        d = M.gen_plaintext(700 + wb, int(rng.integers(0, 4)), n)
        s, _ = M.qtm_encode(d, wb)
        m = sorted(set(int(x) for x in rng.integers(1, n, 300)) |
                   set((k << wb) + j for k in range(1, (n >> wb) + 1) for j in range(-4, 2) if 0 < (k << wb) + j < n))[:2000]
        for what in ("clean", "flip", "cut"):
            b = bytearray(s)
            if what == "flip":
                b[int(rng.integers(len(b) // 4, len(b)))] ^= 1 << int(rng.integers(0, 8))
            if what == "cut":
                b = b[:len(b) * 2 // 3]
            streams.append(bytes(b)); lens.append(n); wbs.append(wb); marks.append(m)
    d = M.gen_plaintext(799, 0, 60000)
    s, _ = M.qtm_encode(d, 12)
    streams.append(bytes(s)); lens.append(d.size); wbs.append(12); marks.append(list(range(1, 20001)))      # (80 KB of log)
    offs, pos = [], 0
    for st in streams:
        pos = (pos + 15) & ~15
        offs.append(pos); pos += len(st)
    tabs = []
    for m in marks:
        pos = (pos + 3) & ~3
        tabs.append(pos); pos += 4 * len(m)
    arena = np.zeros(pos + 64, dtype=np.uint8)
    for st, o in zip(streams, offs):
        arena[o:o + len(st)] = np.frombuffer(st, dtype=np.uint8)
    for m, o in zip(marks, tabs):
        arena[o:o + 4 * len(m)] = np.asarray(m, dtype=np.uint32).view(np.uint8)
    # (room for the logs: 16 bytes of alignment + 4 bytes per mark behind every unit's output)
    units, out_bytes = M.make_units(M.KIND_QUANTUM, offs, [len(st) for st in streams], lens, window_bits=wbs, out_slack=16 + 4 * 20000)
    units["flags"] |= M.UF_QTM_MARKS
    units["in_chunk"] = np.asarray(tabs) // 4
    units["ref_len"] = [len(m) for m in marks]
    out, res = M.decode_batch(units, arena, out_bytes)
    held = fails = 0
    for i, st in enumerate(streams):
        e, want = oracle_qtm_marks(st, lens[i], wbs[i], marks[i])
        _e, o, r = oracle_qtm(st, lens[i], wbs[i])
        assert res["err"][i] == e and res["out_len"][i] == r.out_len, (i, wbs[i], res[i], e, r.out_len)
        lo = int(units["out_off"][i]) + ((lens[i] + 15) & ~15)
        got = out[lo:lo + 4 * len(marks[i])].view(np.uint32).tolist()
        assert got == want, (i, wbs[i], [(p, g, w) for p, g, w in zip(marks[i], got, want) if g != w][:5])
        held += sum(0 < w < M.QTM_MARK_FAILS for w in want); fails += sum(w == M.QTM_MARK_FAILS for w in want)
        if e == 0:
            assert out[int(units["out_off"][i]):int(units["out_off"][i]) + lens[i]].tobytes() == o
    assert held > 1000 and fails > 20, (held, fails)


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
