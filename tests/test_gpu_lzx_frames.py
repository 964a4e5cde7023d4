"""Frame-level parse parallelism of the LZX path (MSPACK_HIP_UF_FRAME_TABLE): units that carry the container's
frame table get their frames parsed by one wavefront each; the unit's own wavefront adopts what was parsed from
exactly its state and decodes everything else serially.  Whatever the table says -- right, wrong, garbage -- the
result must be the oracle's: error code, byte count, flags (but the diagnostic FRAMES_ADOPTED bit), in_next and
every byte.  And where the guess holds (one block per frame) the fast path must really have been taken."""
import numpy as np
import pytest

import libmspack_amd as M
from helpers import oracle_lzx

pytestmark = pytest.mark.gpu
ADOPTED = M.F_FRAMES_ADOPTED


def run(streams, params, tabs):
    """streams: bytes; params: (out_len, wb, reset, e8); tabs: per unit a uint32 array (frame offsets) or None"""
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
    units, out_bytes = M.make_units(M.KIND_LZX, offs, [len(s) for s in streams], [p[0] for p in params],
                                    window_bits=[p[1] for p in params], reset_frames=[p[2] for p in params],
                                    e8_base=[p[3] for p in params], frame_tabs=toff)
    for i, t in enumerate(tabs):
        if t is None:
            units["flags"][i] &= ~np.uint32(M.UF_FRAME_TABLE)
    out, res = M.decode_batch(units, arena, out_bytes)
    return units, out, res


def check(streams, params, units, out, res, compare_bytes=True):
    for i, (s, p) in enumerate(zip(streams, params)):
        e, o, r = oracle_lzx(s, p[0], p[1], p[2], length=p[0], e8_base=p[3])
        assert res["err"][i] == e, (i, res[i], e)
        assert (int(res["flags"][i]) & ~ADOPTED) == r.flags, (i, res[i], r.flags)
        assert res["out_len"][i] == r.out_len, (i, res[i], r.out_len)
        assert res["in_next"][i] == r.in_next, (i, res[i], r.in_next)
        if compare_bytes:
            got = out[units["out_off"][i]:units["out_off"][i] + r.out_len].tobytes()
            assert got == o[:r.out_len], "unit %d differs at byte %d" % (
                i, next(k for k in range(len(got)) if got[k] != o[k]))


MODES = [dict(mode=1), dict(mode=2), dict(mode=0), dict(mode=3), dict(mode=4, block_size=20000),
         dict(mode=0, block_size=9999), dict(mode=1, block_size=65536), dict(repeats=0, lazy=0),
         dict(intel_filesize=250000), dict(intel_filesize=12345, e8_base=5000, mode=2)]


@pytest.mark.parametrize("kw", MODES, ids=[str(k) for k in MODES])
def test_frames_vs_oracle(built, kw):
    data = M.gen_plaintext(17, M.TEXT_MIX, 10 * 32768 + 12345)
    streams, params, tabs = [], [], []
    for wb, reset in [(21, 2), (16, 0), (17, 3), (15, 1), (21, 16)]:
        comp, fo = M.lzx_encode(data, wb, reset, M.lzx_opts(**kw))
        e8 = kw.get("e8_base", 0)
        fo = fo.astype(np.int64)
        streams.append(comp.tobytes()); params.append((data.size, wb, reset, e8)); tabs.append(fo[:-1])
        if reset:
            ib = reset * 32768
            for k in range(0, data.size, ib):
                f0 = k // 32768
                f1 = min((k + ib + 32767) // 32768, len(fo) - 1)
                streams.append(comp[int(fo[f0]):].tobytes())
                params.append((min(ib, data.size - k), wb, reset, e8 + k))
                tabs.append(fo[f0:f1] - fo[f0])
    units, out, res = run(streams, params, tabs)
    check(streams, params, units, out, res)
    one_block_per_frame = kw.get("mode", 0) in (0, 1, 2) and kw.get("block_size", 0) == 0
    if one_block_per_frame:
        assert (res["flags"] & ADOPTED).all(), "the frame-parallel path was not taken where its guess holds"
    if "e8_base" not in kw:
        assert res["err"][0] == 0 and np.array_equal(out[units["out_off"][0]:units["out_off"][0] + data.size], data)


def test_wrong_tables_cost_time_not_correctness(built):
    data = M.gen_plaintext(23, M.TEXT_MIX, 6 * 32768)
    comp, fo = M.lzx_encode(data, 21, 2)
    fo = fo.astype(np.int64)[:-1]
    rng = np.random.default_rng(3)
    variants = [fo, fo + 2, fo - 2, np.zeros_like(fo), fo[::-1].copy(), rng.integers(0, comp.size, fo.size),
                fo + 1, np.full_like(fo, comp.size + 1000), np.concatenate([fo[:3], fo[3:] + 4]), fo * 2,
                np.concatenate([fo[:2], [fo[1]], fo[3:]])]
    streams = [comp.tobytes()] * len(variants)
    params = [(data.size, 21, 2, 0)] * len(variants)
    units, out, res = run(streams, params, variants)
    check(streams, params, units, out, res)
    # (the stream ends on a reset point: the reference's look-ahead then runs out of input, MSPACK_ERR_READ after
    # every byte has been produced -- for the right table and for the wrong ones alike)
    assert (res["out_len"] == data.size).all()
    for i in range(len(variants)):
        assert np.array_equal(out[units["out_off"][i]:units["out_off"][i] + data.size], data), i
    assert res["flags"][0] & ADOPTED


def test_damaged_streams_with_tables(built):
    data = M.gen_plaintext(29, M.TEXT_MIX, 4 * 32768)
    rng = np.random.default_rng(9)
    streams, params, tabs = [], [], []
    for wb, reset, kw in [(21, 2, {}), (17, 0, dict(mode=2)), (16, 4, dict(mode=4, block_size=30000))]:
        comp, fo = M.lzx_encode(data, wb, reset, M.lzx_opts(**kw))
        fo = fo.astype(np.int64)[:-1]
        c = comp.tobytes()
        for _ in range(60):
            b = bytearray(c)
            for _k in range(int(rng.integers(1, 4))):
                k = int(rng.integers(0, len(b))); b[k] ^= 1 << int(rng.integers(0, 8))
            streams.append(bytes(b)); params.append((data.size, wb, reset, 0)); tabs.append(fo)
        for cut in (1, 2, 7, 100, len(c) // 3, int(fo[1]), int(fo[2]) + 1, len(c) - 70, len(c) - 5, len(c) - 1):
            streams.append(c[:cut]); params.append((data.size, wb, reset, 0)); tabs.append(fo)
    units, out, res = run(streams, params, tabs)
    # damaged streams may copy from window bytes the reference never wrote (tests/test_gpu_fuzz.py): bytes are
    # compared only where the oracle's own output is the plaintext
    check(streams, params, units, out, res, compare_bytes=False)
    for i, (s, p) in enumerate(zip(streams, params)):
        e, o, r = oracle_lzx(s, p[0], p[1], p[2], length=p[0])
        if o[:r.out_len] == data.tobytes()[:r.out_len]:
            assert out[units["out_off"][i]:units["out_off"][i] + r.out_len].tobytes() == o[:r.out_len], i


def test_headline_batch_with_tables(built):
    """the 4096-interval batch as bench.py runs it: frame tables on, bit-exact, fast path taken everywhere"""
    n, ub = 4096, 65536
    plain, comp, off, ln, tab = M.corpus_lzx_units(0xBA5E11, 0, n, ub, 21, frame_tables=True)
    units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2, frame_tabs=tab)
    out, res = M.decode_batch(units, comp, out_bytes)
    assert (res["err"] == 0).all() and (res["out_len"] == ub).all()
    assert np.array_equal(out[:n * ub], plain)
    assert (res["flags"] & ADOPTED).all()
    for i in (0, 1, 777, n - 1):
        e, o, r = oracle_lzx(comp[int(off[i]):int(off[i]) + int(ln[i]) + 4].tobytes(), ub, 21, 2)
        assert e == 0 and r.in_next == res["in_next"][i] and (int(res["flags"][i]) & ~ADOPTED) == r.flags


FAMILIES = ["TEXT_ENGLISH", "TEXT_RECORDS", "TEXT_BINARY", "TEXT_REPETITIVE", "TEXT_RANDOM"]


@pytest.mark.parametrize("fam", FAMILIES)
def test_frames_other_plaintexts(built, fam):
    """the lane parser on token statistics unlike the headline corpus's: long matches (few, long tokens: a walk
    falls into step late), nearly incompressible data (frames larger than one 8 KiB pass of the parse wave's LDS
    stage, more records than a frame slot holds, stored blocks), a short last frame"""
    data = M.gen_plaintext(41, getattr(M, fam), 7 * 32768 + 777)
    streams, params, tabs = [], [], []
    for wb, reset, kw in [(21, 2, {}), (16, 0, dict(mode=2)), (18, 4, dict(mode=1)), (21, 0, dict(repeats=0, lazy=0))]:
        comp, fo = M.lzx_encode(data, wb, reset, M.lzx_opts(**kw))
        fo = fo.astype(np.int64)
        streams.append(comp.tobytes()); params.append((data.size, wb, reset, 0)); tabs.append(fo[:-1])
        streams.append(comp.tobytes()); params.append((data.size, wb, reset, 0)); tabs.append(fo[:-1] + 64)   # and a wrong table
    units, out, res = run(streams, params, tabs)
    check(streams, params, units, out, res)
    for i in range(len(streams)):
        assert np.array_equal(out[units["out_off"][i]:units["out_off"][i] + data.size], data), i


_LAUNCH_PATHS_CACHE = None


@pytest.mark.parametrize("env", [{}, {"MSPACK_HIP_STREAM_RESOLVE": "0"}, {"MSPACK_HIP_TICKET_ORDER": "0"}, {"MSPACK_HIP_TICKET_ORDER": "1"},
                                 {"MSPACK_HIP_TICKET_ORDER": "2"}, {"MSPACK_HIP_NO_FRAME_PARSE": "1"}],
                         ids=["pipe", "pipe_no_stream", "level_order", "mixed_sections", "unit_major", "serial"])
def test_launch_paths_same_bytes(built, env, tmp_path_factory):
    """shim.hip launch_kind: the shipped default (mspack_lzx_pipe: one dependency-driven launch) and the serial kernel alone
    (MSPACK_HIP_NO_FRAME_PARSE; round 2's header / parse / unit kernels in a row were removed in round 4) --
    same results, on launches smaller than, about and larger than the chip, and on units of three frames; launches with a wave for
    every ticket (300 units of three frames, 1024 of two) take their frames up while they are parsed (lzx_pipe_resolve_stream, round 6) --
    MSPACK_HIP_STREAM_RESOLVE=0 is the same launches without; MSPACK_HIP_TICKET_ORDER forces one ticket order on every launch (the
    shipped rule picks by the launch's shape: every order is correct, a task only waits for earlier tickets).  Every unit
    carries its table; with the pipe every unit must have had all its frames' records adopted.  (Own process: the
    switches are read when the library loads.)"""
    import os, subprocess, sys
    code = r"""
import numpy as np, sys
sys.path.insert(0, 'tests')
import libmspack_amd as M
from helpers import oracle_lzx
ADOPTED = M.F_FRAMES_ADOPTED
def go(n, ub):
    # (the six runs of this test decode the same five corpora: the first one generates them -- ~30 s of the box's CPU -- and leaves them
    # in the session's temporary directory for the others)
    import os
    cache = os.path.join(os.environ["LAUNCH_PATHS_CACHE"], "c_%d_%d.npz" % (n, ub))
    if os.path.exists(cache):
        z = np.load(cache); plain, comp, off, ln, tab = z["plain"], z["comp"], z["off"], z["ln"], z["tab"]
    else:
        plain, comp, off, ln, tab = M.corpus_lzx_units(0xBA5E11, 0, n, ub, 21, frame_tables=True)
        np.savez(cache + ".tmp.npz", plain=plain, comp=comp, off=off, ln=ln, tab=tab); os.replace(cache + ".tmp.npz", cache)
    units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=ub // 32768, frame_tabs=tab)
    out, res = M.decode_batch(units, comp, out_bytes)
    assert (res['err'] == 0).all() and (res['out_len'] == ub).all() and np.array_equal(out[:n * ub], plain)
    for i in (0, n // 2, n - 1):
        e, o, r = oracle_lzx(comp[int(off[i]):int(off[i]) + int(ln[i]) + 4].tobytes(), ub, 21, ub // 32768)
        assert e == 0 and r.in_next == res['in_next'][i] and (int(res['flags'][i]) & ~ADOPTED) == r.flags
    return float(((res['flags'] & ADOPTED) != 0).mean())
print(go(300, 3 * 32768), go(1024, 65536), go(3600, 32768), go(3600, 3 * 32768), go(6144, 32768))
"""
    global _LAUNCH_PATHS_CACHE
    if _LAUNCH_PATHS_CACHE is None:
        _LAUNCH_PATHS_CACHE = str(tmp_path_factory.mktemp("launch_paths"))
    e2 = dict(os.environ, LAUNCH_PATHS_CACHE=_LAUNCH_PATHS_CACHE); e2.pop("MSPACK_HIP_NO_FRAME_PARSE", None); e2.pop("MSPACK_HIP_STREAM_RESOLVE", None); e2.pop("MSPACK_HIP_TICKET_ORDER", None); e2.update(env)
    r = subprocess.run([sys.executable, "-c", code], env=e2, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(__file__)))
    assert r.returncode == 0, r.stderr[-2000:]
    adopted = [float(x) for x in r.stdout.split()[-5:]]
    want = 0.0 if "MSPACK_HIP_NO_FRAME_PARSE" in env else 1.0
    assert all(a == want for a in adopted), r.stdout


@pytest.mark.parametrize("shape", ["one_table_among_900", "900_tables_and_a_few_without"])
def test_units_with_and_without_tables_in_one_batch(built, shape):
    """the host path numbers the units that carry a frame table first and sizes the frame records and
    the record pool for THOSE slots only; units without a table get frame slots behind them.  The map kernels used to write the
    records of every LZX unit -- with many table-less units far past the scratch (hdr control words, other launches' records):
    the unit with the table then silently dropped off the frame-parallel path, or memory was corrupted.  Both mixes, every
    unit against the oracle, and the units with tables must really have gone through mspack_lzx_pipe."""
    big = M.gen_plaintext(4242, M.TEXT_MIX, 3 * 32768)
    streams, params, tabs = [], [], []
    if shape == "one_table_among_900":
        n_small, n_tab = 900, 1
    else:
        n_small, n_tab = 7, 900
    for i in range(n_tab):
        d = big if n_tab == 1 else M.gen_plaintext(5000 + i, i & 3, 32768 + 700 * (i % 40))
        lz, fo = M.lzx_encode(d, 17, 0)
        streams.append(lz.tobytes()); params.append((d.size, 17, 0, 0)); tabs.append(np.asarray(fo[:-1], dtype=np.uint32))
    for i in range(n_small):
        d = M.gen_plaintext(9000 + i, i & 3, 1500 + 37 * (i % 50))
        lz, _fo = M.lzx_encode(d, 16, 0)
        streams.append(lz.tobytes()); params.append((d.size, 16, 0, 0)); tabs.append(None)
    # (the units with tables in the middle of the list: arena order is not slot order)
    perm = np.random.default_rng(3).permutation(len(streams))
    streams = [streams[j] for j in perm]; params = [params[j] for j in perm]; tabs = [tabs[j] for j in perm]
    units, out, res = run(streams, params, tabs)
    check(streams, params, units, out, res)
    with_tab = np.array([t is not None for t in tabs])
    assert ((res["flags"][with_tab] & ADOPTED) != 0).all(), "a unit with a frame table did not take the frame-parallel path"
    assert ((res["flags"][~with_tab] & ADOPTED) == 0).all()


def test_blocks_that_span_frames(built):
    """Round 5: a frame need not be one block.  Block sizes of several frames, of a fraction of a frame, ending anywhere inside a
    frame (this build's encoder with block_size set), verbatim and aligned: with frame tables every such unit must come out of the
    frame-parallel path (FRAMES_ADOPTED) and equal the oracle in error code, flags, in_next and every byte."""
    data = M.gen_plaintext(31, M.TEXT_MIX, 14 * 32768 + 4321)
    streams, params, tabs = [], [], []
    for bs in (100000, 65536, 32768 * 5, 40000, 20000, 9999, 1 << 20):
        for mode in (0, 1, 2):
            for wb, reset in ((21, 0), (17, 4)):
                comp, fo = M.lzx_encode(data, wb, reset, M.lzx_opts(mode=mode, block_size=bs))
                fo = fo.astype(np.int64)
                if reset == 0:
                    streams.append(comp.tobytes()); params.append((data.size, wb, 0, 0)); tabs.append(fo[:-1])
                else:
                    ib = reset * 32768
                    for k in range(0, data.size, ib):
                        f0, f1 = k // 32768, min((k + ib + 32767) // 32768, len(fo) - 1)
                        streams.append(comp[int(fo[f0]):].tobytes()); params.append((min(ib, data.size - k), wb, reset, k))
                        tabs.append(fo[f0:f1] - fo[f0])
    units, out, res = run(streams, params, tabs)
    check(streams, params, units, out, res)
    assert (res["flags"] & ADOPTED).all(), np.nonzero((res["flags"] & ADOPTED) == 0)[0][:10]


def test_real_cabinet_blocks_of_megabytes(built):
    """The reference's large-files.cab (Microsoft's encoder): LZX-15 and LZX-21 folders whose blocks are MEGABYTES long (the first
    one 8 384 624 bytes: it ends inside frame 255) -- every frame but one in 256 lies inside a block and has no header.  The first
    300 CFDATA blocks of both folders with the cabinet's block sizes as the frame table, against the oracle; the frame-parallel
    path must have taken them (rounds 2-4 sent every real cabinet down the serial path)."""
    import base64, json, os
    import helpers
    kat = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_folders.json")
    k = [v for v in json.load(open(kat)) if v["source"].endswith("large-files-cab.cab")][0]
    e, inner, _r = oracle_lzx(base64.b64decode(k["stream_b64"]), k["out_len"], k["window_bits"], 0)
    assert e == 0
    streams, params, tabs = [], [], []
    for f in helpers.cab_folders(helpers.cab_cut_folders(inner, 300))[1:3]:
        assert (f["comp_type"] & 0x0F) == 3
        blocks = [p for p, _u in f["blocks"]]
        streams.append(b"".join(blocks)); params.append((sum(u for _p, u in f["blocks"]), (f["comp_type"] >> 8) & 0x1F, 0, 0))
        tabs.append(np.cumsum([0] + [len(b) for b in blocks[:-1]]).astype(np.int64))
    units, out, res = run(streams, params, tabs)
    check(streams, params, units, out, res)
    assert (res["flags"] & ADOPTED).all()
