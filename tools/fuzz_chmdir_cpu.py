"""scratch: randomized differential fuzz of the CHM driver's DIRECTORY code (libmspack_amd/csrc/host/chmd.c: headers, PMGL listing,
fast_find through the PMGI index -- host logic only, no decoding) against the REAL reference chmd (oracle/_ref).  A synthetic CHM with a
few hundred to a few thousand entries (tests/chmdir_recipe.py names; PMGI index of one or two levels) gets 1..4 random edits in its
ITSF / ITSP headers and in its directory chunks (byte flips, small integers written over ENCINTs / quickref slots / chunk headers, a
chunk's signature swapped, the file cut inside the directory); compared per CHM: open() error and the listing (name, section, offset,
length), fast open() error and ~120 fast_find answers (error, section, offset, length).  Needs the development container.
    python tools/fuzz_chmdir_cpu.py <seed> [cases]"""
import os, struct, sys
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests"))
import libmspack_amd as M
from libmspack_amd import api
import helpers
import chmdir_recipe as R

_BASE = {}


def base(n):
    if n not in _BASE:
        names = R.synthetic_names(12000)[:: 12000 // n][:n]
        data = M.gen_plaintext(0, 7, 65536)
        lz, fo = M.lzx_encode(data, 16, 2)
        rng = np.random.RandomState(n)
        files = [(nm, int(rng.randint(0, 60000)), int(rng.randint(0, 5000))) for nm in names]
        _BASE[n] = (bytes(M.chm_write(lz, fo, data.size, 16, 2, files)), names)
    return _BASE[n]


def mutate(chm, rng):
    b = bytearray(chm)
    dir_off = 0x78 + 0x54                                            # (chm_write: ITSF 0x60 + two header-section entries... = 0x78; ITSP 0x54)
    chunk_size, = struct.unpack_from("<I", b, 0x78 + 0x10)
    n_chunks, = struct.unpack_from("<I", b, 0x78 + 0x2C)
    dir_end = dir_off + chunk_size * n_chunks
    for _ in range(int(rng.integers(1, 5))):
        r = rng.random()
        if r < .15:   b[int(rng.integers(0, dir_off))] = int(rng.integers(0, 256))                                   # headers
        elif r < .30: struct.pack_into("<I", b, 0x78 + 4 * int(rng.integers(0, 0x15)), int(rng.choice([0, 1, 2, 3, 7, 0xFFFFFFFF, n_chunks, n_chunks - 1, chunk_size])))
        elif r < .60: b[int(rng.integers(dir_off, dir_end))] ^= 1 << int(rng.integers(0, 8))                          # an entry / ENCINT / name byte
        elif r < .75:                                                                                                # a chunk header field
            c = int(rng.integers(0, n_chunks))
            struct.pack_into("<I", b, dir_off + c * chunk_size + 4 * int(rng.integers(0, 5)), int(rng.choice([0, 1, 5, chunk_size, chunk_size - 1, 0xFFFFFFFF, n_chunks])))
        elif r < .85:                                                                                                # the entry count / a quickref slot
            c = int(rng.integers(0, n_chunks))
            struct.pack_into("<H", b, dir_off + (c + 1) * chunk_size - 2 * int(rng.integers(1, 6)), int(rng.choice([0, 1, 2, 0xFFFF, 300, 4000])))
        elif r < .92:                                                                                                # PMGL <-> PMGI <-> junk
            c = int(rng.integers(0, n_chunks))
            b[dir_off + c * chunk_size: dir_off + c * chunk_size + 4] = [b"PMGL", b"PMGI", b"PMGX"][int(rng.integers(0, 3))]
        else:
            del b[int(rng.integers(dir_off, dir_end)):]
            break
    return bytes(b)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    assert helpers.have_ref()
    helpers.ref().refh_zero_alloc(1)
    rng = np.random.default_rng(seed)
    bad = 0
    for k in range(cases):
        chm0, names = base(int(rng.choice([300, 1500, 4000])))
        chm = mutate(chm0, rng) if k % 20 else chm0
        pick = [names[int(i)] for i in rng.integers(0, len(names), 50)]
        q = pick + [R.flip_case(nm) for nm in pick[:30]] + [nm + b"0" for nm in pick[:10]] + [nm[:-1] for nm in pick[10:20]]
        q += [b"/", b"", b"/zzzz", b"!", b"\x7f", b"::DataSpace/Storage/MSCompressed/Content", b"::DataSpace/NameList", b"/\xf0\x9f\x98\x80", b"\xff\xfe", b"A" * 300]
        e, lst = helpers.ref_chm_list(chm)
        with api.Chm(chm, mem=True) as c:
            mine = [(nm[:127], ln, off, sec) for nm, ln, off, sec in c.files] if c.open_error == 0 else []
            ref = [(f["name"], f["length"], f["offset"], f["section"]) for f in lst]
            if c.open_error != e or mine != ref:
                bad += 1
                d = next((i for i in range(min(len(mine), len(ref))) if mine[i] != ref[i]), min(len(mine), len(ref)))
                print("case %d: open reference %d (%d files) mine %d (%d files), first difference at %d: %s / %s" %
                      (k, e, len(ref), c.open_error, len(mine), d, ref[d:d + 1], mine[d:d + 1]))
                continue
        fe, want = helpers.ref_chm_find(chm, q)
        with api.Chm(chm, mem=True, fast=True) as c:
            if c.open_error != fe:
                bad += 1; print("case %d: fast open reference %d mine %d" % (k, fe, c.open_error)); continue
            if fe: continue
            for nm, w in zip(q, want):
                err, f = c.find(nm)
                g = (err, f.section.contents.id if f is not None else -1, f.offset if f is not None else 0, f.length if f is not None else 0)
                if g != tuple(w):
                    bad += 1; print("case %d: find %r reference %s mine %s" % (k, nm[:60], tuple(w), g)); break
    print("seed %d: %d cases, %d mismatches" % (seed, cases, bad))


if __name__ == "__main__":
    main()
