"""Generates tests/golden/chmdir.json (+ copies the reference's small CHM fixtures to tests/golden/chmdir/).

Runs ONLY in the development container, driving the REAL reference (oracle/_ref).  Two parts:
 (1) the reference's own directory fixtures (libmspack/test/test_files/chmd/*.chm -- chmd_test.c:27-125):
     what open() returns (error, or the file list) and what fast_find() answers for every listed name,
     for case-flipped variants and for names that are not there;
 (2) a synthetic CHM with thousands of directory entries, so that the PMGI index has two levels above the
     PMGL chunks (written by libmspack_amd/csrc/corpus/containers.c): fast_find() answers for a sample of
     names (ASCII, mixed case, UTF-8 two/three/four-byte sequences, absent names).
The recipe of (2) is deterministic; the test rebuilds the same CHM and compares with the recorded answers.
"""
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import helpers  # noqa: E402
import chmdir_recipe as R  # noqa: E402

SRC = "/root/reference/libmspack/test/test_files/chmd/"
FIXTURES = ["cve-2015-4468-namelen-bounds.chm", "cve-2015-4469-namelen-bounds.chm", "cve-2015-4472-namelen-bounds.chm",
            "cve-2018-14679-off-by-one.chm", "cve-2018-14682-unicode-u100.chm", "cve-2019-1010305-name-overread.chm",
            "cve-2018-14680-blank-filenames.chm", "cve-2018-18585-blank-filenames.chm",
            "encints-64bit-offsets.chm", "encints-64bit-lengths.chm", "encints-64bit-both.chm",
            "cve-2017-6419-lzx-negative-spaninfo.chm"]


def main():
    assert helpers.have_ref()
    out = {"fixtures": [], "synthetic": None}
    for name in FIXTURES:
        shutil.copyfile(SRC + name, os.path.join(HERE, "chmdir", name))
        os.chmod(os.path.join(HERE, "chmdir", name), 0o644)
        blob = open(SRC + name, "rb").read()
        err, files = helpers.ref_chm_list(blob)
        ent = dict(file=name, open_err=err,
                   files=[[f["name"].decode("latin-1"), f["section"], f["offset"], f["length"]] for f in files])
        queries = R.fixture_queries([f["name"] for f in files])
        rc, res = helpers.ref_chm_find(blob, queries)
        ent["find_open_err"] = rc
        ent["finds"] = [[q.decode("latin-1"), *r] for q, r in zip(queries, res)]
        out["fixtures"].append(ent)
        print(name, "open", err, "files", len(files), "fast_open", rc, "finds", len(res),
              "found", sum(1 for r in res if r[1] >= 0))
    chm, queries = R.synthetic_chm()
    rc, res = helpers.ref_chm_find(chm, queries)
    assert rc == 0
    lerr, lfiles = helpers.ref_chm_list(chm)
    assert lerr == 0
    out["synthetic"] = dict(chm_md5=hashlib.md5(chm).hexdigest(), n_files=len(lfiles),
                            list_md5=hashlib.md5(repr([(f["name"], f["section"], f["offset"], f["length"]) for f in lfiles]).encode()).hexdigest(),
                            finds=[[q.decode("latin-1"), *r] for q, r in zip(queries, res)])
    print("synthetic: files", len(lfiles), "queries", len(queries), "found", sum(1 for r in res if r[1] >= 0))
    # (3) the same CHM with one PMGL chunk's entry count too large: the reference's bad-ENCINT flag is never cleared (chmd.c:262), so
    # the listing ends in that chunk -- and open() still succeeds ("contents are corrupt", chmd.c:166-172)
    bad = R.damaged_listing_chm()
    derr, dfiles = helpers.ref_chm_list(bad)
    out["damaged_listing"] = dict(chm_md5=hashlib.md5(bad).hexdigest(), open_err=derr, n_files=len(dfiles),
                                  list_md5=hashlib.md5(repr([(f["name"], f["section"], f["offset"], f["length"]) for f in dfiles]).encode()).hexdigest())
    print("damaged listing: open", derr, "files", len(dfiles), "of", len(lfiles))
    assert 0 < len(dfiles) < len(lfiles) // 2
    json.dump(out, open(os.path.join(HERE, "chmdir.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
