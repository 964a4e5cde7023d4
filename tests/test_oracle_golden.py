"""CPU: the oracle (our restatement of lzxd/mszipd/qtmd) against the committed golden vectors that
were produced by the REAL reference codecs on the reference's own fixture cabinets
(tests/golden/kat_folders.json; cabd_test.c:405-520 known answers and must-fail vectors)."""
import base64
import hashlib
import json
import os

import pytest

from helpers import oracle_lzx, oracle_mszip, oracle_qtm

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_folders.json")))


@pytest.mark.parametrize("k", KAT, ids=["%s#%d" % (os.path.basename(k["source"]), k["folder"]) for k in KAT])
def test_oracle_matches_reference_kat(k):
    s = base64.b64decode(k["stream_b64"])
    if k["method"] == 1:
        e, o, r, _ = oracle_mszip(s, k["out_len"])
    elif k["method"] == 2:
        e, o, r = oracle_qtm(s, k["out_len"], k["window_bits"])
    else:
        e, o, r = oracle_lzx(s, k["out_len"], k["window_bits"], 0, length=k["out_len"])
    assert e == k["ref_err"]
    assert r.out_len == k["ref_written"]
    if k["deterministic"]:
        assert hashlib.md5(o[:r.out_len]).hexdigest() == k["ref_md5"]


def test_published_md5s():
    """the three codec outputs of mszip_lzx_qtm.cab as listed in libmspack/test/cabd_test.c:472-478"""
    want = {1: "940cba86658fbceb582faecd2b5975d1", 3: "703474293b614e7110b3eb8ac2762b53",
            2: "98fcfa4962a0f169a3c7fdbcb445cf17"}
    seen = 0
    for k in KAT:
        if k["source"].endswith("test_files/cabd/mszip_lzx_qtm.cab"):
            assert k["ref_md5"] == want[k["method"]]
            seen += 1
    assert seen == 3


def test_cab_checksum_on_reference_cabinets():
    """oracle/cab_oracle.c (cabd_checksum, cabd.c:1462-1479) pinned on DATA: every CFDATA block that carries a checksum in the
    reference's own test cabinets -- Microsoft-made files among them -- stores what the restatement computes: the payload with
    seed 0, then the header's cbData / cbUncomp word with that as the seed (cabd.c:1411-1417).  (Files the reference's tests
    damaged on purpose are skipped when their tables do not parse; at least 20 blocks must check out, with payload lengths of every residue mod 4.)"""
    import glob
    import os
    import helpers
    root = os.path.join(os.path.dirname(__file__), "golden")            # ref_fixtures/ (libmspack's tests), cabsets/ (cabextract's split set)
    good = bad = 0
    odd_lengths = set()
    for f in sorted(glob.glob(os.path.join(root, "**", "*.cab"), recursive=True)):
        cab = open(f, "rb").read()
        if cab[:4] != b"MSCF":
            continue
        try:
            blocks = helpers.cab_blocks_with_checksums(cab)
        except Exception:
            continue
        for csum, hdr4, payload in blocks:
            if csum == 0:
                continue
            if helpers.oracle_cab_checksum(hdr4, helpers.oracle_cab_checksum(payload, 0)) == csum:
                good += 1
                odd_lengths.add(len(payload) & 3)
            else:
                bad += 1
    assert good >= 20 and bad <= good // 5, (good, bad)             # (a few fixtures have deliberately bad blocks)
    assert odd_lengths == {0, 1, 2, 3}, odd_lengths                  # every form of the odd tail was exercised
