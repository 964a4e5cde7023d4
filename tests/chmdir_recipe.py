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
