"""CHM directory and fast_find against what the REAL reference answered
(tests/golden/chmdir.json, made by tests/golden/make_chmdir_golden.py): the reference's own directory
fixtures (chmd_test.c:27-125) and a synthetic CHM whose PMGI index is two levels deep.  Host logic only:
no GPU needed."""
import hashlib
import json
import os
import struct

import pytest

from libmspack_amd import api
import chmdir_recipe as R

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "chmdir.json")))


def _finds(c, queries):
    out = []
    for q in queries:
        err, f = c.find(q)
        out.append([err, f.section.contents.id if f is not None else -1, f.offset if f is not None else 0,
                    f.length if f is not None else 0])
    return out


@pytest.mark.parametrize("fx", G["fixtures"], ids=[f["file"] for f in G["fixtures"]])
def test_reference_fixtures(fx):
    path = os.path.join(HERE, "golden", "chmdir", fx["file"])
    with api.Chm(path) as c:
        assert c.open_error == fx["open_err"]
        # (the reference-side lister reports at most 127 name bytes)
        got = [[nm[:127].decode("latin-1"), sec, off, ln] for nm, ln, off, sec in c.files]
        assert got == fx["files"]
    with api.Chm(path, fast=True) as c:
        assert c.open_error == fx["find_open_err"]
        if fx["find_open_err"]:
            return
        queries = [q[0].encode("latin-1") for q in fx["finds"]]
        assert _finds(c, queries) == [q[1:] for q in fx["finds"]]


def test_pmgi_two_levels():
    chm, queries = R.synthetic_chm()
    s = G["synthetic"]
    assert hashlib.md5(chm).hexdigest() == s["chm_md5"]            # same container the reference saw
    # ITSP header: index depth 3, a valid index root (chm.h:43-60)
    depth, index_root = struct.unpack_from("<II", chm, 0x78 + 0x18)
    n_chunks, = struct.unpack_from("<I", chm, 0x78 + 0x2C)
    assert depth == 3 and index_root < n_chunks
    with api.Chm(chm) as c:
        assert c.open_error == 0 and len(c.files) == s["n_files"]
        lst = [(nm, sec, off, ln) for nm, ln, off, sec in c.files]
        assert hashlib.md5(repr(lst).encode()).hexdigest() == s["list_md5"]
    with api.Chm(chm, fast=True) as c:
        assert [q[0].encode("latin-1") for q in s["finds"]] == queries
        assert _finds(c, queries) == [q[1:] for q in s["finds"]]


def test_listing_after_a_bad_encint():
    """one PMGL chunk's entry count too large: the reference lists nothing behind the first badly encoded integer -- its error flag
    is never cleared (chmd.c:262) -- and still opens the file with what it has (chmd.c:166-172)"""
    chm = R.damaged_listing_chm()
    s = G["damaged_listing"]
    assert hashlib.md5(chm).hexdigest() == s["chm_md5"]
    with api.Chm(chm) as c:
        assert c.open_error == s["open_err"] == 0 and len(c.files) == s["n_files"]
        lst = [(nm, sec, off, ln) for nm, ln, off, sec in c.files]
        assert hashlib.md5(repr(lst).encode()).hexdigest() == s["list_md5"]


@pytest.mark.gpu
def test_fast_find_then_extract_gpu(built):
    """F2 end to end on the hardware: `fast_open` (no file list), `fast_find` of every name through the PMGI / PMGL chunks, and
    `extract` of what it found -- LZX intervals decoded by the HIP kernels -- against what the REAL chmd wrote for the same files
    in the same order (tests/golden/chm_extract.json: clean, E8, damaged-content and short-table CHMs), plus names that are not
    there (same code, no file)."""
    import chm_extract_recipe as X
    vecs = json.load(open(os.path.join(HERE, "golden", "chm_extract.json")))
    tags = ("lzx21-r2", "lzx21-r64", "lzx21-r2-e8", "lzx16-r2-flip@f6", "lzx21-r2-fewentries", "lzx21-r2-cut", "config3-1024-intervals")
    for v in [v for v in vecs if v["tag"] in tags]:
        chm, _d, files = X.build(v["case"])
        run = [r for r in v["runs"] if r["order"] == sorted(r["order"])][0]
        with api.Chm(chm, fast=True, mem=True) as c:
            assert c.open_error == v["open_err"] == 0 and c.files == []        # fast_open reads no file list
            for idx, exp in zip(run["order"], run["results"]):
                name, off, ln = files[idx]
                err, f = c.find(name)
                assert err == 0 and f is not None and (f.offset, f.length, f.section.contents.id) == (off, ln, 1), (v["tag"], name)
                err, data = c.extract_found(f)
                assert err == exp["err"] and len(data) == exp["n"] and hashlib.md5(data).hexdigest() == exp["md5"], (v["tag"], idx, err, exp)
            for missing in (b"/nope.bin", b"/f9999.bin", b"/"):
                err, f = c.find(missing)
                assert err == 0 and f is None
