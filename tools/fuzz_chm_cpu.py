"""scratch: randomized differential fuzz of the CHM driver's host logic (libmspack_amd/csrc/host/chmd.c on the CPU stand-in for the
batch ABI) against the REAL reference chmd (oracle/_ref): recipe CHMs (tests/chm_extract_recipe.py: LZX windows 16..21, reset
intervals of 1..4 frames, 5..12 files that cross reset points) with random damage -- bit flips in the compressed content, edits of
ControlData / ResetTable / SpanInfo fields and of reset-table entries, truncation -- and four extraction orders on ONE decompressor
each; per extract() call: the reference's error code, byte count and bytes.  Needs the development container (oracle/_ref).
    python tools/fuzz_chm_cpu.py <seed> [cases]"""
import ctypes, glob, os, subprocess, sys
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
from libmspack_amd import api
import helpers
import chm_extract_recipe as R
from fuzz_drivers_cpu import hostlogic

def recipe(rng, k):
    rf = int(rng.integers(1, 5)); wb = int(rng.integers(16, 22))
    n_int = int(rng.integers(2, 9))
    n = n_int * rf * R.FRAME - (int(rng.integers(0, R.FRAME)) if rng.random() < 0.5 else 0)
    c = dict(seed=9000 + k, text=int(rng.integers(0, 4)), n_bytes=n, window_bits=wb, reset_frames=rf,
             files=R.spread_files(n, int(rng.integers(5, 13)), k + 1, rf * R.FRAME, pinned=(rf * R.FRAME, n - 1)))
    if rng.random() < 0.3: c["intel_filesize"] = int(rng.integers(1000, n))
    muts = []
    nfr = (n + R.FRAME - 1) // R.FRAME
    for _ in range(int(rng.integers(0, 3))):
        r = rng.random()
        if r < .35: muts.append(["flip_content", int(rng.integers(0, nfr)), int(rng.integers(0, 600)), int(rng.integers(0, 8))])
        elif r < .5: muts.append(["rtable_entry", int(rng.integers(0, nfr)), int(rng.integers(0, 1 << 20))])
        elif r < .6: muts.append(["rtable_u32", int(rng.choice([4, 8, 0x0C, 0x10, 0x18, 0x20])), int(rng.integers(0, 1 << 17))])
        elif r < .7: muts.append(["control_u32", int(rng.choice([8, 0x0C, 0x10])), int(rng.choice([1, 2, 3, 0x8000, 0x10000, 4, 64]))])
        elif r < .8: muts.append(["spaninfo", int(rng.integers(0, 2 * n))])
        elif r < .9: muts.append(["cut", int(rng.integers(1, 3000))])
        else: c["uncomp_len"] = int(rng.integers(1, 2 * n))
    c["mutations"] = muts
    return c

def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    assert helpers.have_ref()
    helpers.ref().refh_zero_alloc(1)
    L = hostlogic()
    rng = np.random.default_rng(seed)
    bad = 0
    for k in range(cases):
        case = recipe(rng, seed * 100000 + k)
        try:
            chm, _d, files = R.build(case)
        except Exception as ex:
            continue
        e, lst = helpers.ref_chm_list(chm)
        with api.Chm(chm, mem=True, L=L) as c:
            if c.open_error != e: bad += 1; print("case %d: open reference %d mine %d  %s" % (k, e, c.open_error, case["mutations"])); continue
        if e: continue
        n = len(lst)
        orders = [list(range(n)), list(range(n - 1, -1, -1)), [int(x) for x in rng.permutation(n)], [n // 2, n // 2, 0, n - 1, n - 1, 1 % n]]
        for order in orders:
            cap = sum(lst[i]["length"] for i in order) + 4096
            if cap > (1 << 27): continue
            rc, want = helpers.ref_chm_extract(chm, order, cap=cap)
            if rc: continue
            with api.Chm(chm, mem=True, L=L) as c:
                for j, (i, (we, wb)) in enumerate(zip(order, want)):
                    c.mem.outputs.clear()
                    ge, gb = c.extract(i)
                    if ge != we or (we == 0 and gb != wb):
                        bad += 1
                        print("case %d order %s call %d (file %d): reference (%d, %d bytes) mine (%d, %d bytes)  rf %d wb %d %s" %
                              (k, order[:6], j, i, we, len(wb), ge, len(gb), case["reset_frames"], case["window_bits"], case["mutations"]))
                        break
    print("seed %d: %d cases, %d mismatches" % (seed, cases, bad))

if __name__ == "__main__":
    main()
