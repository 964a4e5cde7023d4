"""Cabinet sets: append / prepend / search / split CFDATA blocks against what the REAL
reference answered for its own fixtures (tests/golden/cabsets.json, made by tests/golden/make_cabset_golden.py
from cabd_test.c:284-400 and cabextract's split.test / search.test files).

Host logic (joins, merged lists, search, stored folders) runs without a GPU; extraction of the MSZIP
split set is the GPU part."""
import hashlib
import json
import os

import pytest

import libmspack_amd as M
from libmspack_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))
G = json.load(open(os.path.join(HERE, "golden", "cabsets.json")))


def fixture(name):
    return os.path.join(HERE, "golden", "cabsets", name)


def run_ops(s, ops):
    return [(s.prepend if op else s.append)(a, b) for op, a, b in ops]


@pytest.mark.parametrize("sc", G["scenarios"], ids=[s["name"] for s in G["scenarios"]])
def test_join_and_list(sc):
    with api.CabSet([fixture(c) for c in sc["cabs"]]) as s:
        assert s.open_errors == [0] * len(sc["cabs"])
        assert run_ops(s, sc["ops"]) == sc["op_errs"]
        got = s.files(sc["list_cab"])
        want = [(f["name"].encode("latin-1"), f["length"], f["offset"], f["comp_type"], f["folder"], f["folder_blocks"])
                for f in sc["files"]]
        assert got == want
        # every cabinet of one chain shows the same lists (cabd_test.c:382-397)
        c = s.cabs[sc["list_cab"]]
        w = c.contents.prevcab
        while w:
            assert api.C.addressof(w.contents.files.contents) == api.C.addressof(c.contents.files.contents)
            w = w.contents.prevcab


def _extract_all(sc, L=None):
    with api.CabSet([fixture(c) for c in sc["cabs"]], L=L) as s:
        run_ops(s, sc["ops"])
        out = []
        for fp, f in zip(s.file_ptrs(sc["list_cab"]), sc["files"]):
            err, data = s.extract(fp)
            out.append((f["name"], err, len(data), hashlib.md5(data).hexdigest()))
        return out


STORED = [s for s in G["scenarios"] if all(f["comp_type"] == 0 for f in s["files"])]
CODED = [s for s in G["scenarios"] if s not in STORED]


@pytest.mark.parametrize("sc", STORED, ids=[s["name"] for s in STORED])
def test_extract_stored_sets(sc):
    """multi_basic_pt*: one stored folder split over five cabinets -- no codec, so no GPU"""
    assert _extract_all(sc) == [(f["name"], f["err"], f["out_len"], f["md5"]) for f in sc["files"]]


@pytest.mark.gpu
@pytest.mark.parametrize("sc", CODED, ids=[s["name"] for s in CODED])
def test_extract_split_sets(sc):
    """cabextract's split-[1-5].cab (MSZIP folders continued across cabinets, split CFDATA blocks)"""
    assert _extract_all(sc) == [(f["name"], f["err"], f["out_len"], f["md5"]) for f in sc["files"]]


@pytest.mark.parametrize("sc", CODED, ids=[s["name"] for s in CODED])
def test_extract_split_sets_host_logic_cpu(built, hostlogic, sc):
    """the same sets through the same driver code on the CPU stand-in for the batch ABI (tests/csrc/batch_standin.c): split
    blocks whose parts carry their own checksums (verified by checksum units of the batch), chains that run out of cabinets
    (the input is cut at the failing read: such a folder's checksums are verified while it is read, cabd.c: gather_folder)"""
    assert _extract_all(sc, L=hostlogic) == [(f["name"], f["err"], f["out_len"], f["md5"]) for f in sc["files"]]


@pytest.mark.parametrize("se", G["searches"], ids=["%s-%d" % (s["file"], s["searchbuf"]) for s in G["searches"]])
def test_search(se):
    err, found = api.cab_search(fixture(se["file"]), se["searchbuf"])
    assert err == 0
    assert found == [(o, k, nm.encode("latin-1")) for o, k, nm in se["found"]]
