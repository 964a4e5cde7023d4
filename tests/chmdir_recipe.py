"""Deterministic recipes shared by tests/golden/make_chmdir_golden.py (reference side, dev container) and
tests/test_chmdir.py (our side): the query names for the reference's CHM fixtures and the synthetic
many-entry CHM whose PMGI index has two levels."""
import numpy as np

import libmspack_amd as M


def flip_case(b):
    return bytes((c ^ 0x20) if (65 <= c <= 90 or 97 <= c <= 122) else c for c in b)


def fixture_queries(names):
    q = []
    for nm in names:
        q += [nm, flip_case(nm), nm + b"x", nm[:-1] if len(nm) > 1 else b"/"]
    q += [b"/", b"", b"/zzzzzzzz", b"::DataSpace/NameList", b"/\xc3\xa9t\xc3\xa9.html", b"\xff\xfe", b"A" * 300]
    return q


def synthetic_names(n=12000):
    rng = np.random.RandomState(20240917)
    words = [b"alpha", b"Beta", b"GAMMA", b"delta", b"\xc3\xa9psilon", b"Zeta", b"\xce\xb7ta", b"theta",
             b"\xe4\xb8\xad\xe6\x96\x87", b"Iota", b"kappa", b"\xf0\x9f\x98\x80", b"Lambda", b"mu", b"NU", b"xi"]
    names = set()
    while len(names) < n:
        k = rng.randint(1, 4)
        parts = [words[rng.randint(len(words))] + (b"%d" % rng.randint(0, 900)) for _ in range(k)]
        names.add(b"/Documentation/Reference-Manual/" + b"/".join(parts) + (b".htm", b".PNG", b".css", b"")[rng.randint(4)])
    # case-insensitive duplicates would make the directory order ambiguous: keep one per folded name
    seen, outl = set(), []
    for nm in sorted(names):
        f = nm.lower()
        if f not in seen:
            seen.add(f); outl.append(nm)
    return outl


def synthetic_chm():
    """-> (chm bytes, query names).  One tiny LZX stream; every entry points into it."""
    names = synthetic_names()
    data = M.gen_plaintext(0, 7, 65536)
    lz, fo = M.lzx_encode(data, 16, 2)
    rng = np.random.RandomState(7)
    files = []
    for nm in names:
        off = int(rng.randint(0, 60000)); ln = int(rng.randint(0, 5000))
        files.append((nm, off, min(ln, 65536 - off)))
    chm = M.chm_write(lz, fo, data.size, 16, 2, files)
    q = []
    pick = rng.choice(len(names), 400, replace=False)
    for i in pick:
        nm = names[i]
        q += [nm, flip_case(nm)]
    q += [names[0], names[-1], names[len(names) // 2], b"/", b"", b"/zzzz", b"/alpha", b"/documentation/reference-manual/\xce\x97ta1",
          b"/DOCUMENTATION/REFERENCE-MANUAL/\xc3\x89psilon3.htm",
          b"!", b"\x7f", b"/\xf0\x9f\x98\x80", b"::DataSpace/Storage/MSCompressed/Content", b"::dataspace/storage/mscompressed/controldata"]
    q += [nm + b"0" for nm in names[::97]] + [nm[:-1] for nm in names[::101]]
    return chm, q


def damaged_listing_chm():
    """the synthetic CHM with one entry more announced in its fourth PMGL chunk than it holds, made of the chunk's free space and
    quickref area so that its section number runs into the end of the chunk: a badly encoded integer in the middle of the listing
    (tools/fuzz_chmdir_cpu.py, round 4)"""
    import struct
    chm, _ = synthetic_chm()
    b = bytearray(chm)
    dir_off = 0x78 + 0x54
    chunk_size, = struct.unpack_from("<I", b, 0x78 + 0x10)
    c = dir_off + 3 * chunk_size
    n, = struct.unpack_from("<H", b, c + chunk_size - 2)
    assert b[c:c + 4] == b"PMGL" and n > 0
    p = c + 0x14
    for _ in range(n):                                            # name length, name, section, offset, length (chm.h:62-67)
        for field in range(4):
            v = 0
            while True:
                ch = b[p]; p += 1
                v = (v << 7) | (ch & 0x7F)
                if not ch & 0x80:
                    break
            if field == 0:
                p += v
    end = c + chunk_size - 2
    L = end - p - 2                                               # a "name" that leaves one byte in front of the entry count ...
    assert 2 <= L < 128
    struct.pack_into("<H", b, end, n + 1)
    b[p] = L; b[p + 1] = ord("x"); b[p + 2] = ord("y")
    b[end - 1] = 0x80                                             # ... and that byte says the section number goes on
    return bytes(b)
