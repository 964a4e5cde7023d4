"""The libmspack-compatible object API (include/mspack.h) on the GPU decoder, replayed against what
the REAL reference drivers answered (tests/golden/driver_cabs.json, made in the dev container by
tests/golden/make_driver_golden.py): per extract() call the error code, the number of bytes written
and their MD5, for the reference's own fixture cabinets and for corrupted / truncated copies of a
synthetic MSZIP + LZX + Quantum + stored cabinet, in several extraction orders
(cf. libmspack/test/cabd_test.c:405-520).  CHM extraction against the real chmd: tests/test_chm_extract.py."""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

import libmspack_amd as M
from libmspack_amd import api

VECS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "driver_cabs.json")))
BASES = {v["tag"]: base64.b64decode(v["cab_b64"]) for v in VECS if "cab_b64" in v}


def cab_bytes(v):
    if "cab_b64" in v:
        return BASES[v["tag"]]
    b = bytearray(BASES[v["base"]])
    m = v["mutation"]
    if "flip" in m:
        b[m["flip"][0]] ^= 1 << m["flip"][1]
    if "cut" in m:
        b = b[:m["cut"]]
    return bytes(b)


def replay(v, L=None):
    cab = cab_bytes(v)
    p = v["params"]
    for run in (v["runs"] or [None]):
        # the same in-memory mspack_system semantics the goldens were recorded with (api.MemSystem)
        with api.Cab(cab, fix_mszip=p.get("fix_mszip", 0), salvage=p.get("salvage", 0), mem=True, L=L) as c:
            assert c.open_error == v["open_err"], v["tag"]
            if run is None:
                continue
            got_files = [(n.decode("latin1"), ln, off) for n, ln, off, _ct in c.files]
            assert got_files == [(f["name"], f["length"], f["offset"]) for f in v["files"]]
            for idx, exp in zip(run["order"], run["results"]):
                err, data = c.extract(idx)
                tag = "%s file %d (order %s)" % (v["tag"], idx, run["order"])
                assert err == exp["err"], (tag, err, exp)
                if exp["err"] == 0:
                    assert len(data) == exp["n"] and hashlib.md5(data).hexdigest() == exp["md5"], tag


@pytest.mark.gpu
@pytest.mark.parametrize("v", VECS, ids=[v["tag"] for v in VECS])
def test_cab_driver_vs_reference(built, v):
    replay(v)


CPU_VECS = [v for v in VECS if not v["params"].get("fix_mszip")]      # (the stand-in has no MSZIP repair mode: those run on the GPU)


@pytest.mark.parametrize("v", CPU_VECS, ids=[v["tag"] for v in CPU_VECS])
def test_cab_driver_host_logic_cpu(built, hostlogic, v):
    """the same goldens through the same driver code (csrc/host/cabd.c) on the CPU stand-in for the batch ABI (tests/csrc/
    batch_standin.c: the oracle, incl. the feeder's failed reads -- MSPACK_HIP_UF_HARD_EOF): host logic without a GPU"""
    replay(v, L=hostlogic)


@pytest.mark.gpu
def test_cab_any_order_24_permutations(built):
    """cabd_test.c:486-520: every ordering of 4 files from 2 folders gives identical contents."""
    import itertools
    data = M.gen_plaintext(3, 0, 120000)
    lz, fo = M.lzx_encode(data, 18, 0)
    lb = [lz[int(fo[i]):int(fo[i + 1])].tobytes() for i in range(len(fo) - 1)]
    us = [min(32768, data.size - i * 32768) for i in range(len(lb))]
    qs, fs = M.qtm_encode(data, 17)
    pos, qb = 0, []
    for k in fs:
        qb.append(qs[pos:pos + int(k)]); pos += int(k) + 1
    files = [(b"a", 50000, 0, 0), (b"b", 70000, 50000, 0), (b"c", 32768, 0, 1), (b"d", 87232, 32768, 1)]
    cab = M.cab_write([(0x1203, lb, us), (0x1102, qb, us)], files)
    exp = [data[o:o + n].tobytes() for _nm, n, o, _f in files]
    with api.Cab(cab) as c:
        for perm in itertools.permutations(range(4)):
            for i in perm:
                err, d = c.extract(i)
                assert err == 0 and d == exp[i], (perm, i)


@pytest.mark.gpu
def test_chm_driver(built):
    """CHM: LZX-21, reset interval 2 frames; listed order, reverse order, fast_find, interleaved
    (cf. libmspack/test/chmd_order.c:55-129), plus the section-0 system files."""
    n_int = 48
    d = M.gen_plaintext(2, 0, n_int * 65536)
    lz, fo = M.lzx_encode(d, 21, 2)
    rng = np.random.default_rng(4)
    cuts = np.sort(rng.choice(np.arange(1, d.size), size=60, replace=False))
    cuts[10] = 5 * 65536                      # a file boundary exactly on a reset point
    cuts[11] = 5 * 65536 + 32768              # ... and one on a frame boundary
    cuts = np.unique(np.concatenate([[0], cuts, [d.size]]))
    files = [(b"/doc%03d.html" % i, int(cuts[i]), int(cuts[i + 1] - cuts[i])) for i in range(len(cuts) - 1)]
    chm = M.chm_write(lz, fo, d.size, 21, 2, files)
    with api.Chm(chm) as c:
        assert c.open_error == 0
        listed = c.files
        assert [(n, ln, off) for n, ln, off, _s in listed] == [(n, ln, off) for n, off, ln in files]
        for order in (range(len(files)), reversed(range(len(files))), rng.permutation(len(files))):
            for i in order:
                err, out = c.extract(int(i))
                assert err == 0 and out == d[files[i][1]:files[i][1] + files[i][2]].tobytes(), i
    with api.Chm(chm, fast=True) as c:
        assert c.open_error == 0 and c.files == []
        for i in (7, 0, 33, 11):
            err, f = c.find(files[i][0])
            assert err == 0 and f is not None and f.offset == files[i][1] and f.length == files[i][2]
            err, out = c.extract_found(f)
            assert err == 0 and out == d[files[i][1]:files[i][1] + files[i][2]].tobytes()
        err, f = c.find(b"/no-such-file")
        assert err == 0 and f is None
        err, f = c.find(b"::DataSpace/Storage/MSCompressed/ControlData")
        err, out = c.extract_found(f)
        assert err == 0 and len(out) == 28 and out[4:8] == b"LZXC"
