"""Makes tests/golden/cab_qtm_carry.json (development container, oracle/_ref): what the REAL cabd writes when the files of ONE Quantum
folder are extracted and a call fails -- qtmd decodes whole tokens, the match that covers a request's last byte runs past it, and the
NEXT call hands the rest to its output before it decodes anything (qtmd.c:268-276): a call that then fails has still written those
bytes.  And, with a window below the frame size, requests that end inside a match which crosses the window's end, in front of that
end: the reference cannot serve them (qtmd.c:358-374, MSPACK_ERR_DECRUNCH) although the folder decodes.  Cabinets: tests/cab_recipe.py
qtm_cab() -- one folder, ~14 files whose boundaries are chosen with the oracle's marks (half of them where a request holds bytes
back, for the small windows also inside window-crossing matches), undamaged and with a flipped bit in the middle of a file.
    python tests/golden/make_cab_qtm_carry_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import helpers
import cab_recipe as F
import libmspack_amd as M


def main():
    assert helpers.have_ref()
    helpers.ref().refh_zero_alloc(1)
    gold = []
    for seed, wb, n, kind in ((8100, 15, 100000, 0), (8101, 12, 60000, 0), (8102, 10, 30000, 2), (8103, 16, 150000, 1),
                              (8104, 13, 70000, 3), (8105, 21, 120000, 0), (8106, 11, 40000, 0), (8107, 14, 90000, 2)):
        rng = np.random.default_rng(seed)
        data = M.gen_plaintext(seed, kind, n)
        qs, _fs = M.qtm_encode(data, wb)
        # what a request ending at every candidate position holds back (one oracle decode with marks)
        cand = sorted(set(int(x) for x in rng.integers(1, n, 4000)) |
                      set((k << wb) - j for k in range(1, (n >> wb) + 1) for j in range(1, 6) if 0 < (k << wb) - j < n))
        e, log = helpers.oracle_qtm_marks(bytes(qs), n, wb, cand)
        assert e == 0
        held = [p for p, c in zip(cand, log) if 0 < c < 0xFFFFFFFF]
        fails = [p for p, c in zip(cand, log) if c == 0xFFFFFFFF]
        clean = [p for p, c in zip(cand, log) if c == 0]
        pick = lambda v, k: [v[int(i)] for i in rng.choice(len(v), size=min(k, len(v)), replace=False)] if v else []
        cuts = sorted(set(pick(held, 7) + pick(fails, 3) + pick(clean, 3)))
        cab, _ = F.qtm_cab(seed, wb, cuts, n, kind)
        nf = len(cuts) + 1
        for flip_at in (None, "mid"):
            c = bytearray(cab)
            flip = None
            if flip_at:
                # a bit in the middle of the folder's data: the file that holds the damage fails after its skip succeeded
                flip = 36 + 8 + (len(c) - 44) // 2
                c[flip] ^= 0x08
            c = bytes(c)
            v = dict(seed=seed, wb=wb, n=n, kind=kind, cuts=cuts, flip=flip, cab_md5=hashlib.md5(c).hexdigest(),
                     held=sum(p in held for p in cuts), fails=sum(p in fails for p in cuts), runs=[])
            orders = [[i] for i in range(nf)] + [list(range(nf)), list(range(nf - 1, -1, -1)), [nf // 2, nf // 2 + 1, nf // 2 - 1]]
            for salvage in (0, 1):
                for order in orders:
                    order = [o for o in order if 0 <= o < nf]
                    rc, res = helpers.ref_cab_extract(c, order, cap=len(order) * (n + 4096) + 4096, salvage=salvage)
                    assert rc == 0
                    v["runs"].append(dict(salvage=salvage, order=order,
                                          results=[dict(err=e_, n=len(b), md5=hashlib.md5(b).hexdigest()) for e_, b in res]))
            print(seed, wb, "flip" if flip else "clean", "files", nf, "held", v["held"], "fails", v["fails"],
                  "failing calls", sum(r["err"] != 0 for run in v["runs"] for r in run["results"]),
                  "of them with bytes", sum(r["err"] != 0 and r["n"] != 0 for run in v["runs"] for r in run["results"]))
            gold.append(v)
    json.dump(gold, open(os.path.join(HERE, "cab_qtm_carry.json"), "w"), indent=0)


if __name__ == "__main__":
    main()
