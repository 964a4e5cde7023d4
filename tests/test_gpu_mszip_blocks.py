"""Block-level parse parallelism of the MSZIP path (MSPACK_HIP_UF_FRAME_TABLE on MSZIP units): the CFDATA blocks of a
folder are parsed by one wavefront each, the folder's own wavefront commits the tokens and decodes whatever a record
does not cover the serial way.  Whatever the table says, the result must be the oracle's (error code, byte count,
bytes); where nothing is unusual the fast path must really have been taken."""
import zlib

import numpy as np
import pytest

import libmspack_amd as M
from helpers import oracle_mszip
from test_gpu_mszip import ck_block

pytestmark = pytest.mark.gpu
ADOPTED = M.F_FRAMES_ADOPTED


def folder_blocks(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, history=True, bs=32768, stored_every=0):
    """-> (stream, block offsets)"""
    out, offs, prev, pos = [], [], None, 0
    for n, k in enumerate(range(0, len(data), bs)):
        blk = data[k:k + bs]
        lv = 0 if (stored_every and n % stored_every == stored_every - 1) else level
        b = ck_block(blk, lv, strategy, prev if history else None)
        offs.append(pos); out.append(b); pos += len(b)
        prev = data[max(0, k + bs - 32768):k + bs]
    return b"".join(out), offs


def run(streams, out_lens, tabs, flags=0):
    offs, toff, pos = [], [], 0
    for s, t in zip(streams, tabs):
        pos = (pos + 15) & ~15
        offs.append(pos); pos += len(s) + 8
        pos = (pos + 3) & ~3
        toff.append(pos); pos += 4 * (0 if t is None else len(t))
    arena = np.zeros(pos + 64, dtype=np.uint8)
    for s, o, t, to in zip(streams, offs, tabs, toff):
        arena[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
        if t is not None:
            arena[to:to + 4 * len(t)] = np.asarray(t, dtype=np.uint32).view(np.uint8)
    units, out_bytes = M.make_units(M.KIND_MSZIP, offs, [len(s) for s in streams], out_lens, flags=flags,
                                    out_slack=32768, frame_tabs=toff)
    for i, t in enumerate(tabs):
        if t is None:
            units["flags"][i] &= ~np.uint32(M.UF_FRAME_TABLE)
    # (frame_base was laid out before the flags of table-less units were cleared: slots they do not use are harmless)
    out, res = M.decode_batch(units, arena, out_bytes)
    return units, out, res


def check(streams, out_lens, units, out, res, plains=None):
    for i, s in enumerate(streams):
        e, o, r, _bl = oracle_mszip(s, out_lens[i])
        assert res["err"][i] == e, (i, res[i], e)
        assert res["out_len"][i] == r.out_len, (i, res[i], r.out_len)
        got = out[units["out_off"][i]:units["out_off"][i] + r.out_len].tobytes()
        if e == 0 or plains is None:
            assert got == o[:r.out_len], "unit %d differs at %d" % (i, next((k for k in range(len(got)) if got[k] != o[k]), -1))
        if plains is not None and e == 0:
            assert got == plains[i][:r.out_len]


def test_folders_with_tables(built):
    streams, lens, tabs, plains = [], [], [], []
    for seed, kind, nbytes, kw in [(1, M.TEXT_MIX, 8 * 32768, {}), (2, M.TEXT_ENGLISH, 5 * 32768 + 777, {}),
                                   (3, M.TEXT_BINARY, 12 * 32768, dict(level=9)), (4, M.TEXT_RECORDS, 3 * 32768, dict(level=1)),
                                   (5, M.TEXT_MIX, 6 * 32768, dict(strategy=zlib.Z_FIXED)),
                                   (6, M.TEXT_MIX, 6 * 32768, dict(strategy=zlib.Z_HUFFMAN_ONLY)),
                                   (7, M.TEXT_REPETITIVE, 9 * 32768 + 5, {}), (8, M.TEXT_RANDOM, 4 * 32768, {}),
                                   (9, M.TEXT_MIX, 7 * 32768, dict(stored_every=3)), (10, M.TEXT_MIX, 10 * 32768, dict(history=False)),
                                   (11, M.TEXT_MIX, 40 * 32768 + 100, {})]:
        d = M.gen_plaintext(seed, kind, nbytes).tobytes()
        s, offs = folder_blocks(d, **kw)
        streams.append(s); lens.append(len(d)); tabs.append(offs); plains.append(d)
    units, out, res = run(streams, lens, tabs)
    check(streams, lens, units, out, res, plains)
    assert (res["err"] == 0).all()
    # (not adopted by design: Z_HUFFMAN_ONLY blocks hold 32768 literal tokens, more than a parse wave stores; random
    # data makes zlib emit stored deflate blocks, which the parse waves leave to the folder's wave)
    normal = [0, 1, 2, 3, 4, 6, 8, 9, 10]
    assert all(res["flags"][i] & ADOPTED for i in normal), res["flags"]


def test_wrong_tables_and_odd_folders(built):
    d = M.gen_plaintext(21, M.TEXT_MIX, 6 * 32768).tobytes()
    s, offs = folder_blocks(d)
    offs = np.array(offs, dtype=np.int64)
    rng = np.random.default_rng(4)
    variants = [offs, offs + 1, offs - 1, np.zeros_like(offs), offs[::-1].copy(), rng.integers(0, len(s), offs.size),
                np.full_like(offs, len(s) + 500), np.concatenate([offs[:2], offs[3:], offs[-1:]]), None]
    streams = [s] * len(variants); lens = [len(d)] * len(variants)
    units, out, res = run(streams, lens, variants)
    check(streams, lens, units, out, res, [d] * len(variants))
    assert (res["err"] == 0).all() and (res["flags"][0] & ADOPTED)
    # a SHORT block in the middle (the next block's history is then not a full block right below it), partial requests,
    # a folder whose blocks are 20000 bytes
    d2 = M.gen_plaintext(22, M.TEXT_MIX, 200000).tobytes()
    pieces, prev, soffs, pos = [], None, [], 0
    for (a, b) in [(0, 32768), (32768, 40000), (40000, 72768), (72768, 105536), (105536, 138304), (138304, 171072), (171072, 200000)]:
        blk = ck_block(d2[a:b], 6, zlib.Z_DEFAULT_STRATEGY, prev)
        soffs.append(pos); pieces.append(blk); pos += len(blk); prev = d2[max(0, b - 32768):b]
    s2 = b"".join(pieces)
    s3, o3 = folder_blocks(d2, bs=20000)
    streams = [s2, s, s, s3]; lens = [200000, 100000, 32768 * 3, len(d2)]
    units, out, res = run(streams, lens, [soffs, offs, offs, o3])
    check(streams, lens, units, out, res)


def test_damaged_folders_with_tables(built):
    d = M.gen_plaintext(31, M.TEXT_MIX, 5 * 32768).tobytes()
    rng = np.random.default_rng(8)
    streams, lens, tabs = [], [], []
    for kw in (dict(), dict(strategy=zlib.Z_FIXED), dict(stored_every=2)):
        s, offs = folder_blocks(d, **kw)
        for _ in range(50):
            b = bytearray(s)
            for _k in range(int(rng.integers(1, 4))):
                k = int(rng.integers(0, len(b))); b[k] ^= 1 << int(rng.integers(0, 8))
            streams.append(bytes(b)); lens.append(len(d)); tabs.append(offs)
        for cut in (1, 3, 100, offs[1], offs[2] + 1, len(s) // 2, len(s) - 60, len(s) - 3, len(s) - 1):
            streams.append(s[:cut]); lens.append(len(d)); tabs.append(offs)
    units, out, res = run(streams, lens, tabs)
    for i, s in enumerate(streams):
        e, o, r, _bl = oracle_mszip(s, lens[i])
        assert res["err"][i] == e, (i, res[i], e)
        assert res["out_len"][i] == r.out_len, (i, res[i], r.out_len)
        # (damaged streams may copy window bytes the reference never wrote: bytes only where the oracle's are the plaintext)
        if o[:r.out_len] == d[:r.out_len]:
            assert out[units["out_off"][i]:units["out_off"][i] + r.out_len].tobytes() == o[:r.out_len], i


def test_repair_mode_ignores_tables(built):
    d = M.gen_plaintext(41, M.TEXT_MIX, 4 * 32768).tobytes()
    s, offs = folder_blocks(d)
    b = bytearray(s); b[offs[1] + 40] ^= 0x20
    units, out, res = run([bytes(b), s], [len(d)] * 2, [offs, offs], flags=M.UF_MSZIP_REPAIR)
    for i, st in enumerate([bytes(b), s]):
        e, o, r, _bl = oracle_mszip(st, len(d), repair=1)
        assert res["err"][i] == e and res["out_len"][i] == r.out_len and not (res["flags"][i] & ADOPTED)


def test_pipe_launch_same_results(built):
    """MSPACK_HIP_MSZIP_PIPE=1: block parse tasks and folder tasks in ONE launch (mspack_mszip_pipe: hand-off through
    status words between running waves; measured slower than the default parse kernel + folder kernel, kept as an
    experiment) -- the same checks in a fresh process"""
    import os, subprocess, sys
    env = dict(os.environ, MSPACK_HIP_MSZIP_PIPE="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "folders_with_tables or wrong_tables or damaged"], env=env, cwd=root, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=1800)
    assert p.returncode == 0, p.stdout.decode()[-3000:]
