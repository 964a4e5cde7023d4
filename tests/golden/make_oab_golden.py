"""Generates tests/golden/oab.json: what the REAL reference's msoab_decompressor (oracle/_ref, development
container only) answers for the OAB files of tests/oab_recipe.py -- full files and incremental patches made
with our LZX DELTA encoder -- and for damaged copies of them: error code, bytes written, MD5 of the output.
The recipe is deterministic; tests/test_oab.py rebuilds the same files (their MD5s are recorded here)."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import helpers  # noqa: E402
import oab_recipe as R  # noqa: E402

N_DAMAGED = 40


def main():
    assert helpers.have_ref()
    out = []
    for name, blob, base, want in R.cases():
        err, got = helpers.ref_oab(blob, base)
        assert err == 0 and got == want, name
        ent = dict(name=name, blob_md5=hashlib.md5(blob).hexdigest(), out_len=len(want),
                   out_md5=hashlib.md5(want).hexdigest(), damaged=[])
        rng = np.random.default_rng(len(blob))
        for m in R.damaged(blob, rng, N_DAMAGED):
            e, o = helpers.ref_oab(m, base)
            ent["damaged"].append([e, len(o), hashlib.md5(o).hexdigest()])
        # a different decompression buffer changes how much of a truncated stored block is written
        e, o = helpers.ref_oab(blob[:len(blob) * 2 // 3], base, decompbuf=1000)
        ent["truncated_buf1000"] = [e, len(o), hashlib.md5(o).hexdigest()]
        if base is not None:
            e, o = helpers.ref_oab(blob, base[:len(base) // 2])
            ent["short_base"] = [e, len(o), hashlib.md5(o).hexdigest()]
        out.append(ent)
        print(name, "errors among damaged:", sorted(set(d[0] for d in ent["damaged"])))
    json.dump(out, open(os.path.join(HERE, "oab.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
