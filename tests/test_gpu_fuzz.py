"""Differential fuzzing on the GPU: several hundred damaged streams per codec (bit flips, byte and word
overwrites, truncations, spliced-in garbage, cut-and-paste of stream parts) must give the oracle's
error code, byte count and -- up to that count -- bytes.  The speculative paths (64 tokens per round,
queued matches, vectorised block headers) have many hand-over points to the EOF-exact scalar loops;
this is where a damaged stream lands in them at every possible token."""
import zlib

import numpy as np
import pytest

import libmspack_amd as M
from helpers import oracle_lzx, oracle_mszip, oracle_qtm
from test_gpu_lzx import run_units as run_lzx
from test_gpu_mszip import run as run_mszip, folder as zip_folder
from test_gpu_qtm import run as run_qtm

pytestmark = pytest.mark.gpu


def mutations(s, rng, n):
    """n damaged copies of the byte string s"""
    out = []
    L = len(s)
    for i in range(n):
        b = bytearray(s)
        m = i % 7
        if m == 0:                                              # one bit
            k = int(rng.integers(0, L)); b[k] ^= 1 << int(rng.integers(0, 8))
        elif m == 1:                                            # a few bits close together
            k = int(rng.integers(0, L))
            for _ in range(3):
                j = min(L - 1, k + int(rng.integers(0, 6))); b[j] ^= 1 << int(rng.integers(0, 8))
        elif m == 2:                                            # overwrite 1-8 bytes with noise
            k = int(rng.integers(0, L)); w = int(rng.integers(1, 9))
            b[k:k + w] = bytes(rng.integers(0, 256, size=min(w, L - k), dtype=np.uint8))
        elif m == 3:                                            # truncate
            b = b[:int(rng.integers(0, L))]
        elif m == 4:                                            # zero or 0xFF run
            k = int(rng.integers(0, L)); w = int(rng.integers(1, 40))
            b[k:k + w] = bytes([0 if i & 8 else 255]) * min(w, L - k)
        elif m == 5:                                            # delete a slice (everything after it shifts)
            k = int(rng.integers(0, L)); w = int(rng.integers(1, 16))
            del b[k:k + w]
        else:                                                   # copy one part of the stream over another
            k = int(rng.integers(0, L)); j = int(rng.integers(0, L)); w = int(rng.integers(1, 64))
            seg = bytes(b[j:j + w]); b[k:k + len(seg)] = seg
        out.append(bytes(b))
    return out


def compare(kind, i, res, units, out, e, o, r):
    assert res["err"][i] == e, (kind, i, res[i], e)
    assert res["out_len"][i] == r.out_len, (kind, i, res[i], r.out_len)
    got = out[units["out_off"][i]:units["out_off"][i] + r.out_len].tobytes()
    if got != o[:r.out_len]:
        k = next(j for j in range(len(got)) if got[j] != o[j])
        raise AssertionError("%s stream %d: byte %d of %d differs (err %d)" % (kind, i, k, r.out_len, e))


def test_fuzz_lzx(built):
    rng = np.random.default_rng(20240926)
    streams, params = [], []
    cfgs = [(17, 0, dict(mode=4, block_size=12345)), (21, 2, dict()), (16, 1, dict(mode=2)),
            (18, 0, dict(mode=1)), (15, 0, dict(mode=3)), (21, 2, dict(intel_filesize=200000)),
            (19, 4, dict(mode=4, block_size=40001, intel_filesize=5000, e8_base=1000))]
    for ci, (wb, rf, kw) in enumerate(cfgs):
        n = 90000 if rf == 0 else 32768 * max(rf, 1) * 2
        data = M.gen_plaintext(100 + ci, ci % 6, n)
        comp, _ = M.lzx_encode(data, wb, rf, M.lzx_opts(**kw))
        comp = comp.tobytes()
        e8 = kw.get("e8_base", 0)
        for m in [comp] + mutations(comp, rng, 350):
            streams.append(m + b"\0" * 4 if rf else m); params.append((n, wb, rf, e8))
    units, out, res = run_lzx(streams, params)
    bad = 0
    for i, (s, p) in enumerate(zip(streams, params)):
        e, o, r = oracle_lzx(s, p[0], p[1], p[2], length=p[0], e8_base=p[3])
        compare("lzx", i, res, units, out, e, o, r)
        assert res["flags"][i] == r.flags, (i, res[i], r.flags)
        bad += e != 0
    assert bad > len(streams) // 4                               # the damage does reach the decoder


def lzx_sweep(seed, n_cfg=60):
    """Random LZX configurations rather than hand-picked ones: window 15..21, reset interval 0..4 frames,
    every block mode with random block sizes (block ends fall anywhere in a frame, so the parser of
    lzx_run_spec overshoots and rewinds at arbitrary places), E8 on/off, all plaintext families, lengths from
    1 byte up, a shorter request per stream, and 8 damaged copies of each.  -> streams, params"""
    rng = np.random.default_rng(seed)
    streams, params = [], []
    for c in range(n_cfg):
        wb = int(rng.integers(15, 22)); reset = int(rng.choice([0, 0, 1, 2, 3, 4]))
        n = int(rng.integers(1, 200000))
        kw = {}
        m = int(rng.integers(0, 5))
        if m:
            kw["mode"] = m
        if m in (0, 4):
            kw["block_size"] = int(rng.integers(1, 70000))
        if rng.random() < .2:
            kw["intel_filesize"] = int(rng.integers(1, 400000))
        if rng.random() < .2:
            kw["repeats"] = 0
        if rng.random() < .2:
            kw["lazy"] = 0
        data = M.gen_plaintext(1000 * seed + c, int(rng.integers(0, 6)), n)
        try:
            comp = M.lzx_encode(data, wb, reset, M.lzx_opts(**kw))[0].tobytes()
        except M.MspackHipError:
            continue                    # the corpus encoder's output estimate (tiny blocks of noise)
        tail = b"\0" * 4 if reset else b""
        streams.append(comp + tail); params.append((n, wb, reset, 0))
        for mu in mutations(comp, rng, 8):
            streams.append(mu + tail); params.append((n, wb, reset, 0))
        streams.append(comp + tail); params.append((int(rng.integers(0, n + 1)), wb, reset, 0))
    return streams, params


SWEEP_SEEDS = [11, 12]


@pytest.mark.parametrize("seed", SWEEP_SEEDS)
def test_sweep_lzx_random_configs(built, seed):
    """GPU vs oracle on lzx_sweep(); tests/test_oracle_vs_ref.py runs the same streams oracle vs reference."""
    streams, params = lzx_sweep(seed)
    units, out, res = run_lzx(streams, params)
    for i, (st, p) in enumerate(zip(streams, params)):
        e, o, r = oracle_lzx(st, p[0], p[1], p[2], length=p[0], e8_base=p[3])
        compare("lzx-sweep", i, res, units, out, e, o, r)
        assert res["flags"][i] == r.flags, (i, res[i], r.flags)


def test_fuzz_mszip(built):
    rng = np.random.default_rng(777)
    streams, lens = [], []
    for ci, (level, strat, hist, bs) in enumerate([(6, zlib.Z_DEFAULT_STRATEGY, False, 32768), (9, zlib.Z_DEFAULT_STRATEGY, True, 32768),
                                                   (1, zlib.Z_FIXED, True, 32768), (6, zlib.Z_HUFFMAN_ONLY, False, 32768),
                                                   (6, zlib.Z_DEFAULT_STRATEGY, True, 20000), (0, zlib.Z_DEFAULT_STRATEGY, False, 32768),
                                                   (6, zlib.Z_RLE, True, 32768)]):
        data = M.gen_plaintext(300 + ci, ci % 6, 98304 if bs == 32768 else 80000).tobytes()
        s = zip_folder(data, level, strat, history=hist, bs=bs)
        for m in [s] + mutations(s, rng, 350):
            streams.append(m); lens.append(len(data))
    units, out, res = run_mszip(streams, lens)
    bad = 0
    for i, st in enumerate(streams):
        e, o, r, _ = oracle_mszip(st, lens[i])
        compare("mszip", i, res, units, out, e, o, r)
        bad += e != 0
    assert bad > len(streams) // 4


def test_sweep_mszip_random_configs(built):
    """Random deflate levels / strategies / block sizes / plaintext families, a shorter request and 8 damaged
    copies per folder, a third of the units in repair mode with random feeder chunk sizes."""
    rng = np.random.default_rng(21)
    streams, lens, flags, chunks = [], [], [], []
    for c in range(50):
        n = int(rng.integers(1, 250000))
        data = M.gen_plaintext(5000 + c, int(rng.integers(0, 6)), n).tobytes()
        strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED]))
        bs = 32768 if rng.random() < .6 else int(rng.integers(1, 32769))
        hist = bool(rng.random() < .6) and bs == 32768          # history is only well defined after full blocks
        s = zip_folder(data, int(rng.integers(0, 10)), strat, history=hist, bs=bs)
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
    for i, st in enumerate(streams):
        e, o, r, _ = oracle_mszip(st, lens[i], 0 if not flags[i] else (chunks[i] if chunks[i] else 1))
        compare("mszip-sweep", i, res, units, out, e, o, r)


def repair_corpus():
    """damaged MSZIP folders for repair mode (MSCABD_PARAM_FIXMSZIP) and the feeder chunk size of each"""
    rng = np.random.default_rng(5)
    streams, lens, chunks = [], [], []
    for ci, (level, strat, hist, bs) in enumerate([(6, 0, False, 32768), (9, 0, True, 32768), (1, zlib.Z_FIXED, True, 32768),
                                                   (6, 0, True, 20000), (0, 0, False, 32768),
                                                   (6, zlib.Z_HUFFMAN_ONLY, False, 3000)]):
        data = M.gen_plaintext(300 + ci, ci % 6, 32768 * 6 if bs == 32768 else 100000).tobytes()
        s = zip_folder(data, level, strat, history=hist, bs=bs)
        for k, m in enumerate([s] + mutations(s, rng, 300)):
            streams.append(m); lens.append(len(data)); chunks.append((0, 64, 512, 2, 4096, 1000)[k % 6])
    return streams, lens, chunks


def test_fuzz_mszip_repair_mode(built):
    """mszipd repair mode: a failed block is zero-filled and decoding goes on at the next 'CK' -- looked
    for from where the reference's stream struct was left (last STORE_BITS, or the start of the current
    input chunk after a refill), which may be before or after the damage: error, length and every byte."""
    streams, lens, chunks = repair_corpus()
    offs, pos = [], 0
    for s in streams:
        pos = (pos + 15) & ~15
        offs.append(pos); pos += len(s)
    arena = np.zeros(pos + 64, dtype=np.uint8)
    for s, o in zip(streams, offs):
        arena[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
    units, out_bytes = M.make_units(M.KIND_MSZIP, offs, [len(s) for s in streams], lens, flags=M.UF_MSZIP_REPAIR,
                                    out_slack=32768)
    units["in_chunk"] = chunks
    out, res = M.decode_batch(units, arena, out_bytes)
    repaired = 0
    for i, st in enumerate(streams):
        e, o, r, _ = oracle_mszip(st, lens[i], chunks[i] if chunks[i] else 1)
        compare("mszip-repair", i, res, units, out, e, o, r)
        repaired += e == 0 and oracle_mszip(st, lens[i])[0] != 0
    assert repaired > len(streams) // 8


def test_fuzz_qtm(built):
    rng = np.random.default_rng(4242)
    streams, lens, wbs = [], [], []
    for ci, wb in enumerate([16, 21, 10, 13]):
        data = M.gen_plaintext(500 + ci, ci % 6, 100000)
        s, _ = M.qtm_encode(data, wb)
        for m in [s] + mutations(s, rng, 150):
            streams.append(m); lens.append(data.size); wbs.append(wb)
    units, out, res = run_qtm(streams, lens, wbs)
    for i, st in enumerate(streams):
        e, o, r = oracle_qtm(st, lens[i], wbs[i])
        compare("qtm", i, res, units, out, e, o, r)


def test_long_runs_overflow_the_match_queue(built):
    """Back-to-back maximum-length matches: a single round then emits more output than the start-flag ring
    of spec_queue.hpp covers and more matches than the queue holds, which sends it down the resolve-first /
    copy-one-by-one paths (LZX, MSZIP) -- and runs with period 1, 2, 3, 7 exercise the overlapping copies."""
    rng = np.random.default_rng(5)
    parts = []
    for period in (1, 2, 3, 7, 64, 300):
        unit = bytes(rng.integers(0, 256, period, dtype=np.uint8))
        parts.append(unit * (20000 // period))
        parts.append(M.gen_plaintext(period, 0, 3000).tobytes())
    data = np.frombuffer(b"".join(parts), dtype=np.uint8)
    streams, params = [], []
    for wb, rf, kw in [(17, 0, {}), (21, 2, dict(mode=2)), (15, 0, dict(mode=1))]:
        n = data.size if rf == 0 else (data.size // 65536) * 65536
        comp, _ = M.lzx_encode(data[:n], wb, rf, M.lzx_opts(**kw))
        streams.append(comp.tobytes() + b"\0" * 4); params.append((n, wb, rf, 0))
    units, out, res = run_lzx(streams, params)
    for i, (s, p) in enumerate(zip(streams, params)):
        e, o, r = oracle_lzx(s, p[0], p[1], p[2], length=p[0])
        compare("lzx", i, res, units, out, e, o, r)
        assert e == 0 and o == data[:p[0]].tobytes()
    z = zip_folder(data.tobytes(), 9, zlib.Z_DEFAULT_STRATEGY, history=True)
    units, out, res = run_mszip([z], [data.size])
    e, o, r, _ = oracle_mszip(z, data.size)
    compare("mszip", 0, res, units, out, e, o, r)
    assert e == 0 and o == data.tobytes()
