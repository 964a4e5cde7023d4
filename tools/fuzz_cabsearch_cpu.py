"""scratch: randomized differential fuzz of the cabinet driver's search() (libmspack_amd/csrc/host/cabd.c: the signature scanner and
what it does with candidates that do not hold up -- host logic only) against the REAL reference cabd (oracle/_ref).  A file of random
filler with 1..4 small cabinets at random offsets, some of them damaged (header bytes, cut short, sizes that lie), with decoy "MSCF"
signatures and partial signatures in the filler and across search-buffer boundaries; search buffers of 4..64 bytes and the default.
Compared: the error code and (base offset, number of files, first file name) of every cabinet found, in order.
    python tools/fuzz_cabsearch_cpu.py <seed> [cases]"""
import os, struct, sys
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import libmspack_amd as M
from libmspack_amd import api
import helpers


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    assert helpers.have_ref()
    helpers.ref().refh_zero_alloc(1)
    import cab_recipe
    rng = np.random.default_rng(seed)
    pool = []
    for s in range(6):
        c = bytes(cab_recipe.base_cab(9000 + s))
        pool.append(c)
    bad = 0
    for k in range(cases):
        parts = []
        for _ in range(int(rng.integers(1, 5))):
            fill = bytearray(rng.integers(0, 256, int(rng.integers(0, 3000)), dtype=np.uint8).tobytes())
            for _ in range(int(rng.integers(0, 4))):                       # decoys
                if len(fill) > 40:
                    at = int(rng.integers(0, len(fill) - 36))
                    sig = [b"MSCF", b"MSC", b"MS", b"MSCFMSCF", b"MMSCF"][int(rng.integers(0, 5))]
                    fill[at:at + len(sig)] = sig
                    if rng.random() < 0.5:                                    # a decoy with plausible-looking header fields
                        struct.pack_into("<IIIIIBBHH", fill, at + 4, 0, int(rng.integers(0, 5000)), 0, int(rng.integers(0, 200)), 0, 3, 1,
                                         int(rng.integers(0, 3)), int(rng.integers(0, 3)))
            parts.append(bytes(fill))
            c = bytearray(pool[int(rng.integers(0, len(pool)))])
            r = rng.random()
            if r < .15: c[int(rng.integers(4, 60))] = int(rng.integers(0, 256))
            elif r < .25: del c[int(rng.integers(8, len(c))):]
            elif r < .35: struct.pack_into("<I", c, 8, int(rng.choice([0, 10, len(c) - 1, len(c) + 1, len(c) * 2, 0xFFFFFFFF])))      # cbCabinet
            elif r < .42: struct.pack_into("<I", c, 16, int(rng.choice([0, 10, len(c) - 1, len(c) + 1, 0xFFFFFFFF])))                  # coffFiles
            parts.append(bytes(c))
        if rng.random() < 0.5: parts.append(rng.integers(0, 256, int(rng.integers(0, 500)), dtype=np.uint8).tobytes())
        blob = b"".join(parts)
        sb = int(rng.choice([0, 0, 4, 5, 7, 16, 33, 64, 4096]))
        want = helpers.ref_cab_search(blob, sb)
        err, got = api.cab_search(blob, sb)
        w = want if isinstance(want, list) else []
        got = [(o, n, nm[:63]) for o, n, nm in got]                            # (the reference-side lister reports 63 name bytes)
        if [tuple(x) for x in w] != [tuple(x) for x in got] or (not isinstance(want, list)) != (err != 0 and not got):
            bad += 1
            print("case %d (%d bytes, searchbuf %d): reference %s mine (%d) %s" % (k, len(blob), sb, want if not isinstance(want, list) else
                  [(o, n) for o, n, _ in want], err, [(o, n) for o, n, _ in got]))
    print("seed %d: %d cases, %d mismatches" % (seed, cases, bad))


if __name__ == "__main__":
    main()
