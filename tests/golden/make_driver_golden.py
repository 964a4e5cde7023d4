#!/usr/bin/env python3
"""Regenerates tests/golden/driver_cabs.json -- run ONLY in the development container.

Small cabinets (the reference's own fixtures plus corrupted copies of a synthetic three-codec
cabinet) together with what the REAL reference driver (cabd via oracle/_ref) answers for each file:
error code, bytes written, MD5.  The GPU driver tests replay them through mspack.h."""
import base64
import hashlib
import json
import os
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402
import libmspack_amd as M  # noqa: E402
from helpers import ref_cab_list, ref_cab_extract  # noqa: E402

REF = "/root/reference"
vectors = []


bases = {}


def add(tag, cab, orders=None, base=None, mutation=None, **kw):
    """cabinets derived from a base are stored as base + mutation recipe, not as a full copy"""
    e, lst = ref_cab_list(cab)
    v = dict(tag=tag, open_err=e, params=kw, runs=[])
    if base is None:
        v["cab_b64"] = base64.b64encode(cab).decode()
    else:
        v["base"] = base; v["mutation"] = mutation
    if e == 0:
        n = len(lst)
        v["files"] = [dict(name=f["name"].decode("latin1"), length=f["length"], offset=f["offset"]) for f in lst]
        for order in (orders or [list(range(n))]):
            rc, outs = ref_cab_extract(cab, order, **kw)
            v["runs"].append(dict(order=order, results=[dict(err=er, n=len(o), md5=hashlib.md5(o).hexdigest())
                                                        for er, o in outs]))
    vectors.append(v)


# 1. the reference's own fixtures
T = REF + "/libmspack/test/test_files/cabd/"
for name in ["mszip_lzx_qtm.cab", "normal_2files_2folders.cab", "normal_2files_1folder.cab",
             "cve-2010-2800-mszip-infinite-loop.cab", "cve-2014-9556-qtm-infinite-loop.cab",
             "cve-2015-4470-mszip-over-read.cab", "cve-2015-4471-lzx-under-read.cab",
             "cve-2018-18584-qtm-max-size-block.cab", "lzx-main-tree-no-lengths.cab", "lzx-premature-matches.cab",
             "filename-read-violation-2.cab", "filename-read-violation-3.cab", "filename-read-violation-4.cab",
             "cve-2014-9732-folders-segfault.cab", "bad_signature.cab", "bad_nofolders.cab", "bad_nofiles.cab",
             "reserve_HFD.cab", "reserve_---.cab"]:
    cab = open(T + name, "rb").read()
    orders = None
    if name == "normal_2files_2folders.cab":
        orders = [[0, 1], [1, 0], [0, 0, 1, 1], [1, 1, 0]]
    if name == "cve-2014-9732-folders-segfault.cab":
        orders = [[0, 1, 0]]
    add("ref:" + name, cab, orders)

# 2. a synthetic three-codec cabinet and corrupted copies of it
data = M.gen_plaintext(1, 0, 90000)


def mszip_blocks(d):
    out, us, prev = [], [], None
    for k in range(0, len(d), 32768):
        b = d[k:k + 32768].tobytes()
        c = zlib.compressobj(6, zlib.DEFLATED, -15, 9, 0, prev) if prev else zlib.compressobj(6, zlib.DEFLATED, -15)
        out.append(b"CK" + c.compress(b) + c.flush()); us.append(len(b)); prev = b
    return out, us


mb, mu = mszip_blocks(data)
lz, fo = M.lzx_encode(data, 16, 0)
lb = [lz[int(fo[i]):int(fo[i + 1])].tobytes() for i in range(len(fo) - 1)]
qs, fs = M.qtm_encode(data, 15)
pos, qb = 0, []
for k in fs:
    qb.append(qs[pos:pos + int(k)]); pos += int(k) + 1
us = [min(32768, len(data) - i * 32768) for i in range(len(lb))]
stored = [data[i:i + 20000].tobytes() for i in range(0, 60000, 20000)]
files = [(b"m1.txt", 40000, 0, 0), (b"m2.txt", 50000, 40000, 0), (b"l1.txt", 32768, 0, 1), (b"l2.txt", 57232, 32768, 1),
         (b"q1.txt", 65536, 0, 2), (b"q2.txt", 24464, 65536, 2), (b"s1.bin", 60000, 0, 3)]
cab = M.cab_write([(1, mb, mu), (0x1003, lb, us), (0x0F02, qb, us), (0, stored, [20000] * 3)], files)
orders = [[0, 1, 2, 3, 4, 5, 6], [6, 5, 4, 3, 2, 1, 0], [1, 1, 0, 3, 2, 5, 4]]
add("syn:clean", cab, orders)
SYN = "syn:clean"
rng = np.random.default_rng(11)
for t in range(24):
    b = bytearray(cab)
    k = int(rng.integers(0x24 + 4 * 8 + 120, len(b)))          # somewhere in the data area
    bit = int(rng.integers(0, 8))
    b[k] ^= 1 << bit
    add("syn:flip@%d" % k, bytes(b), orders[:2], base=SYN, mutation=dict(flip=[k, bit]))
    if t % 6 == 0:
        add("syn:flip@%d+fix" % k, bytes(b), orders[:1], base=SYN, mutation=dict(flip=[k, bit]), fix_mszip=1)
        add("syn:flip@%d+salvage" % k, bytes(b), orders[:1], base=SYN, mutation=dict(flip=[k, bit]), salvage=1)
for cut in (len(cab) - 1, len(cab) - 5000, len(cab) // 2, 400):
    add("syn:cut@%d" % cut, cab[:cut], orders[:1], base=SYN, mutation=dict(cut=cut))
json.dump(vectors, open(os.path.join(HERE, "driver_cabs.json"), "w"))
print("wrote", len(vectors), "cabinets,", os.path.getsize(os.path.join(HERE, "driver_cabs.json")), "bytes")
