"""scratch: randomized differential fuzz of the CABINET driver's host logic (libmspack_amd/csrc/host/cabd.c on the CPU stand-in for the
batch ABI, tests/_build/libhostlogic_cpu.so) against the REAL reference driver (oracle/_ref): a synthetic four-folder cabinet (MSZIP with
history, LZX, Quantum, stored) with random byte damage, truncation and header edits; for every file, in two extraction orders and in
the plain and salvage modes: the reference's error code and every byte.  Needs the development container (oracle/_ref).
LIMIT of the stand-in (test infrastructure): no MSZIP repair mode -- that path is covered on the GPU by tests/test_gpu_drivers.py.
What this fuzz found in round 4: DESIGN.md section 7 (extract() after a failed extract(), what a failing Quantum call leaves behind:
fixed, tests/test_cab_sticky.py; two salvage-mode differences understood and left: qtmd's match tail in front of a failure, MSZIP
blocks whose header lies about their uncompressed size).
    python tools/fuzz_drivers_cpu.py <seed> [cases]"""
import ctypes, glob, os, subprocess, sys, zlib
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import libmspack_amd as M
from libmspack_amd import api
import helpers

def hostlogic():
    bdir = os.path.join(R, "tests", "_build"); os.makedirs(bdir, exist_ok=True)
    so = os.path.join(bdir, "libhostlogic_cpu.so")
    srcs = sorted(glob.glob(os.path.join(R, "libmspack_amd", "csrc", "host", "*.c"))) + [os.path.join(R, "tests", "csrc", "batch_standin.c")] + \
        sorted(glob.glob(os.path.join(R, "oracle", "*_oracle.c")))
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-comment", "-I", os.path.join(R, "include"), "-o", so] + srcs + ["-lpthread"])
    return ctypes.CDLL(so)

from cab_recipe import base_cab          # (tests/cab_recipe.py: the four-folder cabinet)

def mutate(cab, rng):
    c = bytearray(cab)
    kind = int(rng.integers(0, 6))
    if kind == 0:                                        # flip a few bytes in the data area
        for _ in range(int(rng.integers(1, 4))):
            c[int(rng.integers(60, len(c)))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:                                      # truncate
        del c[int(rng.integers(40, len(c))):]
    elif kind == 2:                                      # a header / folder / file table byte
        c[int(rng.integers(8, 200))] = int(rng.integers(0, 256))
    elif kind == 3:                                      # a CFDATA header field somewhere: find "CK" and damage the 8 bytes before
        i = bytes(c).find(b"CK", int(rng.integers(60, len(c))))
        if i > 8: c[i - 1 - int(rng.integers(0, 8))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 4:                                      # zero a stretch
        a = int(rng.integers(60, len(c))); n = min(int(rng.integers(1, 300)), len(c) - a); c[a:a + n] = bytes(n)
    return bytes(c)

LAST = []
def mine(L, cab, order, **kw):
    with api.Cab(cab, mem=True, L=L, **kw) as c:
        if c.open_error: return c.open_error, []
        r = []
        for i in order:
            c.mem.outputs.clear()               # (a call that fails before it opens its output leaves the previous one)
            r.append(c.extract(i) if i < len(c.files) else (None, b""))
        LAST[:] = [m for m in c.mem.messages if b"GPU" in (m if isinstance(m, bytes) else m.encode())]
        L.mspack_hip_last_error.restype = ctypes.c_char_p
        if LAST: LAST.append(L.mspack_hip_last_error())
        return 0, r

def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    assert helpers.have_ref()
    helpers.ref().refh_zero_alloc(1)       # (window bytes a damaged stream reads before it wrote them: zero, as on the device)
    L = hostlogic()
    rng = np.random.default_rng(seed)
    bad = 0
    for k in range(cases):
        if k % 25 == 0: cab0 = base_cab(seed * 1000 + k)
        cab = mutate(cab0, rng) if k % 10 else bytes(cab0)
        e, lst = helpers.ref_cab_list(cab)
        for kw in (dict(), dict(salvage=1)):            # (fix-MSZIP mode needs the kernels: the stand-in has no repair mode)
            n = len(lst) if e == 0 else 0
            for order in ([list(range(n)), list(range(n - 1, -1, -1))] if n else [[]]):
                rc, want = helpers.ref_cab_extract(cab, order, cap=(n + 1) * 160000 + 4096, **kw) if e == 0 else (e, [])
                me, got = mine(L, cab, order, **kw)
                if e != 0 or rc != 0:
                    if kw.get("salvage"): continue                       # (the harness lists without salvage mode)
                    if (me != 0) != True: bad += 1; print("case %d %s: reference open/extract rc %d/%d, mine open %d" % (k, kw, e, rc, me))
                    continue
                if me != 0: bad += 1; print("case %d %s: mine open error %d, reference opens" % (k, kw, me)); continue
                for i, ((we, wb_), (ge, gb)) in enumerate(zip(want, got)):
                    if we != ge or (we == 0 and wb_ != gb) or (we != 0 and kw.get("salvage") and wb_ != gb):
                        bad += 1; print("case %d %s order %s file %d: reference (%d, %d bytes) mine (%s, %d bytes)" % (k, kw, order[:3], order[i], we, len(wb_), ge, len(gb)), LAST[-1:]); break
    print("seed %d: %d cases, %d mismatches" % (seed, cases, bad))

if __name__ == "__main__":
    main()
