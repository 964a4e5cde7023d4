"""GPU parity for MSZIP: zlib-made "CK" folders (stored / fixed / dynamic blocks, cross-block history,
short blocks in the middle of a folder, truncation and corruption) vs. the CPU oracle, bit-exact."""
import zlib

import numpy as np
import pytest

import libmspack_amd as M
from helpers import oracle_mszip

pytestmark = pytest.mark.gpu


def ck_block(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, zdict=None):
    if zdict:
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy, zdict)
    else:
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    return b"CK" + c.compress(data) + c.flush()


def folder(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, history=False, bs=32768):
    out, prev = [], None
    for k in range(0, len(data), bs):
        blk = data[k:k + bs]
        out.append(ck_block(blk, level, strategy, prev if history else None))
        prev = data[max(0, k + bs - 32768):k + bs]
    return b"".join(out)


def run(streams, out_lens, flags=0):
    offs, pos = [], 0
    for s in streams:
        pos = (pos + 15) & ~15
        offs.append(pos); pos += len(s)
    arena = np.zeros(pos + 64, dtype=np.uint8)
    for s, o in zip(streams, offs):
        arena[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
    units, out_bytes = M.make_units(M.KIND_MSZIP, offs, [len(s) for s in streams], out_lens, flags=flags,
                                    out_slack=32768)
    out, res = M.decode_batch(units, arena, out_bytes)
    return units, out, res


def check(streams, out_lens, units, out, res, plains=None):
    for i, s in enumerate(streams):
        e, o, r, _bl = oracle_mszip(s, out_lens[i])
        assert res["err"][i] == e, (i, res[i], e)
        assert res["out_len"][i] == r.out_len, (i, res[i], r.out_len)
        assert res["in_next"][i] == r.in_next, (i, res[i], r.in_next)        # (what the last block inflated to beyond the request)
        got = out[units["out_off"][i]:units["out_off"][i] + r.out_len].tobytes()
        if e == 0 or plains is None:
            assert got == o[:r.out_len], "unit %d differs" % i
        if plains is not None and e == 0:
            assert got == plains[i][:r.out_len]


def test_mszip_block_types(built):
    data = M.gen_plaintext(11, M.TEXT_MIX, 200000).tobytes()
    rnd = M.gen_plaintext(12, M.TEXT_RANDOM, 70000).tobytes()
    rep = M.gen_plaintext(13, M.TEXT_REPETITIVE, 100000).tobytes()
    streams, plains = [], []
    for lvl, strat in [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                       (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)]:
        for src in (data, rnd, rep):
            for hist in (False, True):
                streams.append(folder(src, lvl, strat, hist)); plains.append(src)
    lens = [len(p) for p in plains]
    units, out, res = run(streams, lens)
    check(streams, lens, units, out, res, plains)
    assert (res["err"] == 0).all()


def test_mszip_short_blocks_history(built):
    """Short blocks in the middle of a folder: the window restarts at index 0 for every block, so a
    distance that reaches past the current block lands in whatever older block last wrote that
    index (mszipd.c:267-268).  That is not a plain "previous 32 KiB" dictionary -- the expected bytes
    are the reference semantics as restated by the oracle, not the zlib plaintext."""
    data = M.gen_plaintext(11, M.TEXT_MIX, 200000).tobytes()
    streams = [folder(data, 6, history=True, bs=10000), folder(data, 6, history=True, bs=32768 - 5)]
    mixed = ck_block(data[:32768]) + ck_block(data[32768:40000], zdict=data[:32768]) + \
        ck_block(data[40000:72768], zdict=data[7232:40000]) + ck_block(data[72768:80000], zdict=data[40000:72768])
    streams.append(mixed)
    lens = [len(data), len(data), 80000]
    units, out, res = run(streams, lens)
    check(streams, lens, units, out, res, None)
    assert (res["err"] == 0).all()


def test_mszip_partial_request_and_errors(built):
    data = M.gen_plaintext(21, M.TEXT_MIX, 100000).tobytes()
    s = folder(data, 6, history=True)
    rng = np.random.default_rng(3)
    streams, lens = [], []
    for want in (1, 100, 32768, 32769, 65536, 99999, 100000):
        streams.append(s); lens.append(want)
    for cut in (0, 1, 2, 3, 10, 100, len(s) // 2, len(s) - 2, len(s) - 1):
        streams.append(s[:cut]); lens.append(len(data))
    for _ in range(40):
        b = bytearray(s); k = int(rng.integers(0, len(b))); b[k] ^= 1 << int(rng.integers(0, 8))
        streams.append(bytes(b)); lens.append(len(data))
    units, out, res = run(streams, lens)
    for i, st in enumerate(streams):
        e, o, r, _ = oracle_mszip(st, lens[i])
        assert res["err"][i] == e, (i, res[i], e)
        assert res["out_len"][i] == r.out_len, (i, res[i], r.out_len)
        if e == 0 and o == data[:lens[i]]:
            assert out[units["out_off"][i]:units["out_off"][i] + lens[i]].tobytes() == o


def test_mszip_batch_4096_blocks(built):
    """BASELINE config 2: 4096 independent 32 KiB CFDATA blocks."""
    n = 4096
    plain = M.gen_plaintext(77, M.TEXT_MIX, n * 32768)
    blocks = [ck_block(plain[i * 32768:(i + 1) * 32768].tobytes()) for i in range(n)]
    lens = [32768] * n
    units, out, res = run(blocks, lens)
    assert (res["err"] == 0).all() and (res["out_len"] == 32768).all()
    for i in range(0, n, 97):
        o = units["out_off"][i]
        assert np.array_equal(out[o:o + 32768], plain[i * 32768:(i + 1) * 32768])
    # checksum-of-everything property
    total = sum(int(out[units["out_off"][i]:units["out_off"][i] + 32768].astype(np.uint64).sum()) for i in range(n))
    assert total == int(plain.astype(np.uint64).sum())


def test_mszip_request_that_ends_inside_a_block(built):
    """mszipd sizes a block by its deflate stream and keeps what a call did not ask for (mszipd.c:386-392, 440-452): a unit asked for
    fewer bytes than its last block holds says how many more there are (mspack_hip_result.in_next) and leaves them in its slack --
    what the cabinet driver needs when a CFDATA header's uncompressed size is too small.  With block tables too."""
    data = M.gen_plaintext(21, M.TEXT_MIX, 32768 * 2 + 20000).tobytes()
    s = folder(data, history=True)
    asks = [len(data), len(data) - 1, len(data) - 19999, 32768 * 2, 32768 + 5, 1]
    units, out, res = run([s] * len(asks), asks)
    check([s] * len(asks), asks, units, out, res, plains=[data] * len(asks))
    for i, a in enumerate(asks):
        more = int(res["in_next"][i])
        assert more == ((len(data) - a) if a > 32768 * 2 else (32768 - a % 32768) % 32768 if a % 32768 else 0), (i, a, more)
        o = int(units["out_off"][i])
        assert out[o + a:o + a + more].tobytes() == data[a:a + more], (i, a, more)
