"""Runs in the match resolver (spec_queue.hpp: spq_is_run / spq_fill_run): a batch of matches that
follow each other without a literal and share one offset is written as a periodic fill -- no memory round trip per 256 bytes.
What long stretches of zeros and repeated records look like after LZ77 with a maximal match of 257 / 258 bytes; the
reference's large-files.test (a 64-byte line, 2 GiB) is the pure case (tests/test_gpu_large_files.py holds its real folders).
Here: periods below, at and above the wave width (1, 2, 3, 63, 64, 65, 300, 5000, 40000), runs that start and end anywhere,
that cross frames / blocks, with ordinary data before, between and behind them -- LZX units with frame tables (the pipe's
resolve tasks), the same without (serial commit), MSZIP folders with block tables and without.  Every byte against the
plaintext, error codes / flags / in_next against the oracle."""
import zlib

import numpy as np
import pytest

import libmspack_amd as M
from helpers import oracle_lzx, oracle_mszip
from test_gpu_lzx_frames import run as lzx_run, check as lzx_check

pytestmark = pytest.mark.gpu
PERIODS = [1, 2, 3, 63, 64, 65, 300, 5000, 40000]


def periodic_plain(seed, n):
    """ordinary text, then stretches of every period in PERIODS (each 3 000 - 90 000 bytes long, starting wherever the last one
    ended), ordinary text in between"""
    rng = np.random.default_rng(seed)
    parts = [M.gen_plaintext(seed, 0, 3000 + int(rng.integers(0, 500)))]
    total = parts[0].size
    while total < n:
        p = PERIODS[int(rng.integers(0, len(PERIODS)))]
        pat = rng.integers(0, 256, p, dtype=np.uint8) if p > 1 else np.zeros(1, np.uint8)
        ln = int(rng.integers(3000, 90000))
        parts.append(np.tile(pat, ln // p + 2)[:ln])
        parts.append(M.gen_plaintext(seed + total, int(rng.integers(0, 3)), int(rng.integers(1, 2500))))
        total += ln + parts[-1].size
    return np.concatenate(parts)[:n]


@pytest.mark.parametrize("with_tables", [True, False], ids=["frame-tables", "serial"])
def test_lzx_runs_vs_oracle(built, with_tables):
    streams, params, tabs = [], [], []
    for seed, wb, reset in [(1, 21, 0), (2, 17, 0), (3, 21, 2), (4, 16, 4), (5, 15, 0)]:
        data = periodic_plain(seed, 12 * 32768 + 777 * seed)
        comp, fo = M.lzx_encode(data, wb, reset)
        if reset:
            data = data[:(data.size // (reset * 32768)) * reset * 32768]      # whole intervals: one unit per interval below
            comp, fo = M.lzx_encode(data, wb, reset)
        fo = fo.astype(np.int64)
        if reset == 0:
            streams.append(comp.tobytes()); params.append((data.size, wb, 0, 0)); tabs.append(fo[:-1] if with_tables else None)
        else:
            ib = reset * 32768
            for k in range(0, data.size, ib):
                f0, f1 = k // 32768, (k + ib) // 32768
                streams.append(comp[int(fo[f0]):].tobytes()); params.append((ib, wb, reset, 0))
                tabs.append(fo[f0:f1] - fo[f0] if with_tables else None)
    units, out, res = lzx_run(streams, params, tabs)
    lzx_check(streams, params, units, out, res)


@pytest.mark.parametrize("with_tables", [True, False], ids=["block-tables", "serial"])
def test_mszip_runs_vs_oracle(built, with_tables):
    folders = []
    for seed in (11, 12, 13):
        data = periodic_plain(seed, 9 * 32768 + 1000 * seed)
        blocks, prev = [], None
        for k in range(0, data.size, 32768):
            b = data[k:k + 32768].tobytes()
            c = zlib.compressobj(9, zlib.DEFLATED, -15, 9, 0, prev) if prev else zlib.compressobj(9, zlib.DEFLATED, -15)
            blocks.append(b"CK" + c.compress(b) + c.flush()); prev = b
        folders.append((data, blocks))
    offs, lens, toff, pos = [], [], [], 0
    chunks = []
    for data, blocks in folders:
        pos = (pos + 15) & ~15
        s = b"".join(blocks)
        offs.append(pos); lens.append(len(s)); chunks.append((pos, s)); pos += len(s) + 8
        pos = (pos + 3) & ~3
        t = np.cumsum([0] + [len(b) for b in blocks[:-1]]).astype(np.uint32)
        toff.append(pos); chunks.append((pos, t.tobytes())); pos += 4 * len(t)
    arena = np.zeros(pos + 64, dtype=np.uint8)
    for o, b in chunks:
        arena[o:o + len(b)] = np.frombuffer(b, dtype=np.uint8)
    units, out_bytes = M.make_units(M.KIND_MSZIP, offs, lens, [d.size for d, _ in folders], out_slack=32768,
                                    frame_tabs=toff if with_tables else None)
    out, res = M.decode_batch(units, arena, out_bytes)
    for i, (data, blocks) in enumerate(folders):
        e, o, r, _ = oracle_mszip(b"".join(blocks), data.size)
        assert e == 0 and o == data.tobytes()
        assert res["err"][i] == 0 and res["out_len"][i] == data.size, res[i]
        oo = int(units["out_off"][i])
        assert np.array_equal(out[oo:oo + data.size], data), "folder %d differs at byte %d" % (
            i, int(np.nonzero(out[oo:oo + data.size] != data)[0][0]))
