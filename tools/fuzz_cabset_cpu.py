"""scratch: randomized differential fuzz of the cabinet driver's append() / prepend() (libmspack_amd/csrc/host/cabd.c: joining cabinets,
merging the folders and file lists that continue across them, and the stored-folder extraction of what results -- host logic only)
against the REAL reference cabd (oracle/_ref).  The reference's five-part stored set (tests/golden/cabsets/multi_basic_pt1..5.cab, data
files of its own tests), each part with 0..2 random byte edits, joined by 2..7 random operations (either call, any pair of parts incl.
the same one twice, already joined ones and NULL); compared: every operation's return code, the merged file list of a random part
(name, length, offset, compression type, folder ordinal, folder blocks) and every listed file's extract() code and bytes.
With a third argument "split-%d.cab": cabextract's five-part MSZIP set whose CFDATA blocks are split across the parts, decoded through
the CPU stand-in for the batch ABI.
    python tools/fuzz_cabset_cpu.py <seed> [cases] [split-%d.cab]"""
import os, sys
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
from libmspack_amd import api
import helpers


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    assert helpers.have_ref()
    helpers.ref().refh_zero_alloc(1)
    which = sys.argv[3] if len(sys.argv) > 3 else "multi_basic_pt%d.cab"          # or "split-%d.cab": cabextract's MSZIP set with split blocks
    parts = [open(os.path.join(R_, "tests", "golden", "cabsets", which % i), "rb").read() for i in range(1, 6)]
    L = None
    if "split" in which:
        from fuzz_drivers_cpu import hostlogic
        L = hostlogic()                                                     # (MSZIP: the drivers on the CPU stand-in for the batch ABI)
    rng = np.random.default_rng(seed)
    bad = 0
    for k in range(cases):
        blobs = []
        for p in parts:
            b = bytearray(p)
            if k % 8:
                for _ in range(int(rng.choice([0, 0, 1, 2]))):
                    at = int(rng.integers(4, len(b) if rng.random() < 0.3 else min(len(b), 400)))       # (mostly the headers and tables)
                    b[at] = int(rng.choice([0, 1, 2, 0xFD, 0xFE, 0xFF, b[at] ^ (1 << int(rng.integers(0, 8)))]))
            blobs.append(bytes(b))
        if rng.random() < 0.6:                                             # mostly: the right chain in a random order of joins, plus noise
            pairs = [(i, i + 1) for i in range(4)]
            rng.shuffle(pairs)
            ops = [(int(rng.integers(0, 2)), a, b) for a, b in pairs]
            ops = [(op, a, b) if op == 0 else (1, b, a) for op, a, b in ops]
            for _ in range(int(rng.integers(0, 3))):
                ops.insert(int(rng.integers(0, len(ops) + 1)), (int(rng.integers(0, 2)), int(rng.integers(-1, 5)), int(rng.integers(-1, 5))))
        else:
            ops = [(int(rng.integers(0, 2)), int(rng.integers(-1, 5)), int(rng.integers(-1, 5))) for _ in range(int(rng.integers(2, 8)))]
        lc = int(rng.integers(0, 5))
        n, want_ops, want = helpers.ref_cabset(blobs, ops, lc)
        if n < 0:                                                          # a part does not open: same code from open()?
            with api.CabSet(blobs, mem=True, L=L) as s:
                if not any(e == -n for e in s.open_errors):
                    bad += 1; print("case %d: open reference %d mine %s" % (k, -n, s.open_errors))
            continue
        with api.CabSet(blobs, mem=True, L=L) as s:
            if s.open_errors != [0] * 5:
                bad += 1; print("case %d: open reference ok mine %s" % (k, s.open_errors)); continue
            got_ops = [(s.prepend if op else s.append)(a if a >= 0 else None, b if b >= 0 else None) for op, a, b in ops]
            if got_ops != want_ops:
                bad += 1; print("case %d: ops %s reference %s mine %s" % (k, ops, want_ops, got_ops)); continue
            got = s.files(lc)
            w = [(f["name"], f["length"], f["offset"], f["comp_type"], f["folder"] if f["comp_type"] >= 0 else -1, f["folder_blocks"]) for f in want]
            g = [(nm[:63], ln, off, ct, fid, fb) for nm, ln, off, ct, fid, fb in got]
            if g != w:
                bad += 1; print("case %d: list after %s\n   reference %s\n   mine      %s" % (k, ops, w, g)); continue
            for fp, f in zip(s.file_ptrs(lc), want):
                err, data = s.extract(fp)
                if err != f["err"] or data != f["data"]:
                    bad += 1; print("case %d: extract %r reference (%d, %d bytes) mine (%d, %d bytes)" % (k, f["name"], f["err"], len(f["data"]), err, len(data))); break
    print("seed %d: %d cases, %d mismatches" % (seed, cases, bad))


if __name__ == "__main__":
    main()
