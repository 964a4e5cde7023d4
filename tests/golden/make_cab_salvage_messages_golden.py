"""Makes tests/golden/cab_salvage_messages.json (development container, oracle/_ref): WHEN the real cabd says "WARNING; bad block checksum
found" in salvage mode -- in the extract() call whose decoding makes it read the block (cabd.c:1408-1421), and never from a decompressor
that sits in its error state.  The cabinets of tests/golden/cab_sticky.json with a flipped bit in the second block of their Quantum / LZX /
MSZIP folder (seeds 7100, 7125, 7150), four extraction orders each: per call, the number of such lines.
    python tests/golden/make_cab_salvage_messages_golden.py"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import helpers
import cab_recipe as F


def main():
    assert helpers.have_ref()
    helpers.ref().refh_zero_alloc(1)
    sticky = json.load(open(os.path.join(HERE, "cab_sticky.json")))
    gold = []
    for v in sticky:
        if v["seed"] not in (7100, 7125, 7150):
            continue
        cab = F.base_cab(v["seed"]); cab[v["flip"]] ^= 0x10; cab = bytes(cab)
        assert hashlib.md5(cab).hexdigest() == v["cab_md5"]
        g = dict(seed=v["seed"], flip=v["flip"], cab_md5=v["cab_md5"], runs=[])
        for order in (list(range(8)), list(range(7, -1, -1)), [1, 0, 3, 2, 5, 4], [0, 2, 4, 1, 3, 5, 5, 4]):
            helpers.ref_messages()
            rc, res = helpers.ref_cab_extract(cab, order, cap=len(order) * 160000 + 4096, salvage=1)
            assert rc == 0
            per, hnd = [], []
            lines = helpers.ref_messages()
            for l, h in zip(lines, helpers.ref_message_handles()):
                if l.startswith("#extract"): per.append(0); hnd.append("")
                elif per and "bad block checksum" in l: per[-1] += 1; hnd[-1] += h
            # handles: per call, one character per warning -- 'H' = said with the cabinet's file handle (cabd.c:1415), '-' = with NULL
            g["runs"].append(dict(order=order, errs=[e for e, _ in res], warnings=per, handles=hnd))
            print(v["seed"], order, [e for e, _ in res], per, hnd)
        gold.append(g)
    json.dump(gold, open(os.path.join(HERE, "cab_salvage_messages.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
