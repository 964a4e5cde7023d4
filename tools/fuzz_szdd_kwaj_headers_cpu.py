"""scratch: randomized differential fuzz of the SZDD / KWAJ drivers' open() (libmspack_amd/csrc/host/szddd.c, kwajd.c: signatures, the
KWAJ header's optional fields and names -- host logic only, nothing is decoded) against the REAL reference (oracle/_ref): recipe files
(tests/szdd_kwaj_recipe.py) with 1..3 edits in their first 80 bytes or cut short there; compared: open()'s error code and, when it opens,
compression type / format, length, file name (SZDD: the missing character).
    python tools/fuzz_szdd_kwaj_headers_cpu.py <seed> [cases]"""
import os, sys, tempfile
import ctypes as C
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
from libmspack_amd import api
import helpers
import szdd_kwaj_recipe as R


def open_only(kind, blob):
    L = api.lib()
    T = api.MsszddDecompressor if kind == 0 else api.MskwajDecompressor
    create = L.mspack_create_szdd_decompressor if kind == 0 else L.mspack_create_kwaj_decompressor
    destroy = L.mspack_destroy_szdd_decompressor if kind == 0 else L.mspack_destroy_kwaj_decompressor
    create.restype = api._P(T); create.argtypes = [C.c_void_p]; destroy.argtypes = [api._P(T)]
    mem = api.MemSystem(L)                       # (the reference side reads from memory too: there a seek beyond the end FAILS)
    mem.files[b"mem:in"] = bytes(blob)
    d = create(mem.ptr())
    try:
        h = d.contents.open(d, b"mem:in")
        if not h:
            return (d.contents.last_error(d), -1, -1, b"")
        hc = h.contents
        r = (0, hc.format if kind == 0 else hc.comp_type, hc.length, hc.missing_char.rstrip(b"\0") if kind == 0 else (hc.filename or b""))
        d.contents.close(d, h)
        return r
    finally:
        destroy(d)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    assert helpers.have_ref()
    helpers.ref().refh_zero_alloc(1)
    t = R.texts()
    base = [(0, R.szdd_file(t[0][:3000])), (0, R.szdd_file(t[0][:3000], qbasic=True))]
    for m in range(5):
        base.append((1, R.kwaj_file(t[1][:3000], m, name=b"setup", ext=b"ex_")))
    base.append((1, R.kwaj_file(t[2][:3000], 3, name=b"readme", lzh_types=(1, 2, 0, 3, 1), extra=b"extra text")))
    base.append((1, R.kwaj_file(t[0][:3000], 4, length=False)))
    base.append((1, R.kwaj_file(t[3][:3000], 2, ext=b"txt")))
    rng = np.random.default_rng(seed)
    bad = 0
    for k in range(cases):
        kind, blob = base[int(rng.integers(0, len(base)))]
        b = bytearray(blob)
        for _ in range(int(rng.integers(1, 4))):
            r = rng.random()
            top = min(len(b), 80)
            if top < 2: break
            if r < .6: b[int(rng.integers(0, top))] = int(rng.choice([0, 1, 0xFF, 0x80, 0x41, int(rng.integers(0, 256))]))
            elif r < .8: b[int(rng.integers(0, top))] ^= 1 << int(rng.integers(0, 8))
            else: del b[int(rng.integers(1, top)):]
        b = bytes(b)
        w = helpers.ref_szdd_kwaj(kind, b, cap=1 << 16)
        want = (w["open_err"], w["comp_type"], w["length"], w["filename"]) if w["open_err"] == 0 else (w["open_err"], -1, -1, b"")
        got = open_only(kind, b)
        if want != got:
            bad += 1; print("case %d kind %d: reference %s mine %s   %s" % (k, kind, want, got, b[:40].hex()))
    print("seed %d: %d cases, %d mismatches" % (seed, cases, bad))


if __name__ == "__main__":
    main()
