"""Generates tests/golden/szdd_kwaj.json: what the REAL reference (oracle/_ref, development container) answers
for the SZDD and KWAJ files of tests/szdd_kwaj_recipe.py and for damaged copies of them: open error, extract
error, header fields, output length and MD5.  The recipe is deterministic; tests/test_szdd_kwaj.py rebuilds
the same files (their MD5s are recorded)."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import helpers  # noqa: E402
import szdd_kwaj_recipe as R  # noqa: E402
from test_szdd_kwaj import file_cases, N_DAMAGED, sig, damaged_files  # noqa: E402


def main():
    assert helpers.have_ref()
    out = []
    for name, kind, blob, want in file_cases():
        r = helpers.ref_szdd_kwaj(kind, blob)
        assert r["open_err"] == 0 and r["err"] == 0 and r["data"] == want, name
        ent = dict(name=name, blob_md5=hashlib.md5(blob).hexdigest(), ok=sig(r), damaged=[])
        for m in damaged_files(name, kind, blob, N_DAMAGED):
            # a damaged stream may copy from window bytes the reference never initialised: ask three times,
            # with other work in between; keep only what it always agrees on (None = not comparable)
            ra = helpers.ref_szdd_kwaj(kind, m)
            a = sig(ra)
            # MSZIP keeps its window in uninitialised heap memory: a damaged stream that reaches into bytes no
            # block ever wrote copies whatever the heap held; only outputs that are still a
            # prefix of the plaintext are comparable byte for byte
            if kind == 1 and blob[8] == 4 and ra["data"] != want[:len(ra["data"])]:
                a[6] = None
            helpers.ref_szdd_kwaj(kind, blob)
            b = sig(helpers.ref_szdd_kwaj(kind, m))
            helpers.ref_szdd_kwaj(1, file_cases()[8][2])
            c = sig(helpers.ref_szdd_kwaj(kind, m))
            if a[:6] != b[:6] or a[:6] != c[:6]:
                a = None
            elif a[6] is not None and (a[6] != b[6] or a[6] != c[6]):
                a[6] = None
            ent["damaged"].append(a)
        out.append(ent)
        print(name, "errs among damaged:", sorted(set((d[0], d[1]) for d in ent["damaged"] if d)),
              "unstable:", sum(1 for d in ent["damaged"] if d is None), "unstable bytes:", sum(1 for d in ent["damaged"] if d and d[6] is None))
    json.dump(out, open(os.path.join(HERE, "szdd_kwaj.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
