"""Generates tests/golden/cabsets.json (+ copies the cabinet fixtures into tests/golden/cabsets/).

Runs ONLY in the development container: it drives the REAL reference (oracle/_ref, built from
/root/reference by `make -C oracle ref`) over the reference's own multi-cabinet fixtures
   libmspack/test/test_files/cabd/multi_basic_pt[1-5].cab, search_basic.cab, search_tricky1.cab
   cabextract/test/cabs/split-[1-5].cab, search.cab
through append()/prepend()/search()/extract(), and records what it answers: the return code of every
join, the merged file list, and error code + length + MD5 of every extracted file.  The fixtures are
data files of the reference's tests (cabd_test.c:284-400, cabextract/test/split.test, search.test);
split.test's MD5s are re-checked here.
"""
import ctypes as C
import hashlib
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402

L = "/root/reference/libmspack/test/test_files/cabd/"
X = "/root/reference/cabextract/test/cabs/"
FIX = {**{"multi_basic_pt%d.cab" % i: L + "multi_basic_pt%d.cab" % i for i in range(1, 6)},
       **{"split-%d.cab" % i: X + "split-%d.cab" % i for i in range(1, 6)},
       "search_basic.cab": L + "search_basic.cab", "search_tricky1.cab": L + "search_tricky1.cab",
       "search.cab": X + "search.cab"}

APP, PRE = 0, 1
SPLIT = ["split-%d.cab" % i for i in range(1, 6)]
MULTI = ["multi_basic_pt%d.cab" % i for i in range(1, 6)]
SCENARIOS = [
    dict(name="split_append_in_order", cabs=SPLIT, ops=[(APP, 0, 1), (APP, 1, 2), (APP, 2, 3), (APP, 3, 4)], list_cab=0),
    dict(name="split_prepend_backwards", cabs=SPLIT, ops=[(PRE, 4, 3), (PRE, 3, 2), (PRE, 2, 1), (PRE, 1, 0)], list_cab=4),
    dict(name="split_haphazard", cabs=SPLIT, ops=[(APP, 0, 1), (PRE, 2, 1), (APP, 3, 4), (PRE, 3, 2)], list_cab=2),
    dict(name="split_pairs_then_join", cabs=SPLIT, ops=[(APP, 3, 4), (APP, 1, 2), (APP, 2, 3), (APP, 0, 1)], list_cab=0),
    dict(name="split_first_three_only", cabs=SPLIT[:3], ops=[(APP, 0, 1), (APP, 1, 2)], list_cab=0),
    dict(name="split_last_three_only", cabs=SPLIT[2:], ops=[(APP, 0, 1), (APP, 1, 2)], list_cab=0),
    dict(name="split_2_alone", cabs=SPLIT[1:2], ops=[], list_cab=0),
    dict(name="split_1_alone", cabs=SPLIT[:1], ops=[], list_cab=0),
    dict(name="split_gap", cabs=[SPLIT[0], SPLIT[2]], ops=[(APP, 0, 1)], list_cab=0),
    dict(name="split_wrong_order", cabs=[SPLIT[1], SPLIT[0]], ops=[(APP, 0, 1)], list_cab=0),
    dict(name="merge_args", cabs=MULTI[:2],
         ops=[(APP, 0, -1), (APP, -1, 0), (APP, 0, 0), (PRE, 0, -1), (PRE, -1, 0), (PRE, 0, 0),
              (APP, 0, 1), (APP, 1, 0), (PRE, 0, 1), (PRE, 1, 0), (APP, 0, 1)], list_cab=0),
    dict(name="multi_haphazard", cabs=MULTI, ops=[(APP, 0, 1), (PRE, 2, 1), (APP, 3, 4), (PRE, 3, 2)], list_cab=0),
    dict(name="multi_in_order", cabs=MULTI, ops=[(APP, 0, 1), (APP, 1, 2), (APP, 2, 3), (APP, 3, 4)], list_cab=4),
    dict(name="multi_circular", cabs=MULTI[:3], ops=[(APP, 0, 1), (APP, 1, 2), (APP, 2, 0), (PRE, 0, 2)], list_cab=1),
    dict(name="multi_mixed_sets", cabs=[MULTI[0], SPLIT[1]], ops=[(APP, 0, 1)], list_cab=0),
]
SEARCHES = [("search_basic.cab", 0), ("search_tricky1.cab", 0), ("search.cab", 0), ("search.cab", 4), ("search.cab", 7),
            ("search_basic.cab", 5), ("split-1.cab", 0)]
SPLIT_MD5 = {b"small1.bin": "2ad5ba0f497f1e597ab187a2dfaa2e29", b"small2.bin": "1f862f9e36a32a74202c1120b9f06af7",
             b"medium1.bin": "0a7bd124a4c03a30329bd9ff06f71df7", b"medium2.bin": "b4b0a02ad6a1170d4b3db18cec616fcc",
             b"small3.bin": "bbaecacfeba976165e9d77bbecb0cbde", b"medium3.bin": "b98fe17e8afbcf05aefc5b2c4badbc28"}


def run_set(blobs, ops, list_cab):
    return helpers.ref_cabset(blobs, ops, list_cab)


def main():
    assert helpers.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    os.makedirs(os.path.join(HERE, "cabsets"), exist_ok=True)
    data = {}
    for name, src in FIX.items():
        shutil.copyfile(src, os.path.join(HERE, "cabsets", name))
        os.chmod(os.path.join(HERE, "cabsets", name), 0o644)
        data[name] = open(src, "rb").read()
    out = {"scenarios": [], "searches": []}
    for sc in SCENARIOS:
        n, op_errs, files = run_set([data[c] for c in sc["cabs"]], sc["ops"], sc["list_cab"])
        ent = dict(name=sc["name"], cabs=sc["cabs"], ops=[list(o) for o in sc["ops"]], list_cab=sc["list_cab"],
                   n_files=n, op_errs=op_errs,
                   files=[dict(name=f["name"].decode("latin-1"), length=f["length"], offset=f["offset"],
                               comp_type=f["comp_type"], folder=f["folder"], folder_blocks=f["folder_blocks"],
                               err=f["err"], out_len=len(f["data"]), md5=hashlib.md5(f["data"]).hexdigest())
                          for f in files])
        out["scenarios"].append(ent)
        if sc["name"] in ("split_append_in_order", "split_prepend_backwards", "split_haphazard", "split_pairs_then_join"):
            got = {f["name"]: hashlib.md5(f["data"]).hexdigest() for f in files}
            assert got == SPLIT_MD5, (sc["name"], got)        # cabextract/test/split.test
            assert all(f["err"] == 0 for f in files)
        print(sc["name"], "ops", op_errs, "files", [(f["name"], f["err"], len(f["data"])) for f in files])
    for fname, sbuf in SEARCHES:
        res = helpers.ref_cab_search(data[fname], sbuf)
        out["searches"].append(dict(file=fname, searchbuf=sbuf, found=[[o, k, nm.decode("latin-1")] for o, k, nm in res]))
        print("search", fname, sbuf, res)
    json.dump(out, open(os.path.join(HERE, "cabsets.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
