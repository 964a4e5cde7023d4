#!/usr/bin/env python3
"""Regenerates tests/golden/chm_extract.json -- run ONLY in the development container.

What the REAL reference chmd (oracle/_ref, built from /root/reference by oracle/Makefile) answers for
every extract() call on synthetic CHMs: error code, bytes written, MD5 -- per call, for several call
orders on ONE decompressor each (the lifetime of the reference's lzxd instance, chmd.c:989-1041, decides
where decoding restarts, which damaged intervals a request crosses and the origin of the E8 translation,
lzxd.c:712).  Cases: LZX-21/16, reset intervals of 2 and 64 frames, intel_filesize != 0, a damaged
interval, files across reset points, backwards / repeated requests, dishonest UncompLen, broken reset
tables (-> SpanInfo fallback, chmd.c:1159-1166), reset-table entries beyond the content, truncated
content, a file longer than the section, and BASELINE config 3 (1024 intervals).
The CHMs themselves are not stored: tests/chm_extract_recipe.py rebuilds them from the recipe."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402
from helpers import ref_chm_list, ref_chm_extract  # noqa: E402
import chm_extract_recipe as R  # noqa: E402

I64 = 65536
cases = []


def orders_for(n, seed, extra=()):
    rng = np.random.default_rng(seed)
    o = [list(range(n)), list(range(n - 1, -1, -1)), [int(x) for x in rng.permutation(n)]]
    mid = n // 2
    o.append([mid, mid + 1, mid, 0, n - 1, n - 1, 1][:max(1, min(7, n))] if n > 2 else [0])
    return o + [list(e) for e in extra]


def add(tag, case, orders):
    chm, _d, files = R.build(case)
    e, lst = ref_chm_list(chm)
    v = dict(tag=tag, case=case, chm_md5=hashlib.md5(chm).hexdigest(), open_err=e, runs=[])
    if e == 0:
        # the reference lists files in directory order; the recipe's file table is the same list
        assert [(f["name"], f["offset"], f["length"]) for f in lst] == [(n, o, l) for n, o, l in files], tag
        for order in orders:
            cap = sum(files[i][2] for i in order) + 4096
            rc, outs = ref_chm_extract(chm, order, cap=cap)
            assert rc == 0
            v["runs"].append(dict(order=order, results=[dict(err=er, n=len(o), md5=hashlib.md5(o).hexdigest())
                                                        for er, o in outs]))
    cases.append(v)
    print("%-28s open %d, %d runs, errs %s" % (tag, e, len(v["runs"]),
                                                 sorted({r["err"] for run in v["runs"] for r in run["results"]})))


def base(n_int, wb=21, rf=2, seed=7, text=0, n_files=14, **kw):
    n = n_int * rf * R.FRAME
    c = dict(seed=seed, text=text, n_bytes=n, window_bits=wb, reset_frames=rf,
             files=R.spread_files(n, n_files, seed + 1, rf * R.FRAME,
                                  pinned=(3 * rf * R.FRAME, 3 * rf * R.FRAME + R.FRAME, n - rf * R.FRAME)))
    c.update(kw)
    return c


# 1. clean streams: windows 21 / 16, reset 2 frames; files tile the stream, some end on reset points / frame ends
c = base(12)
add("lzx21-r2", c, orders_for(len(c["files"]), 1))
c = base(12, wb=16, seed=9)
add("lzx16-r2", c, orders_for(len(c["files"]), 2))
# 2. reset interval = window = 64 frames (mspack.h:1525-1530): far offsets, few reset points
c = base(2, wb=21, rf=64, seed=11, n_files=10)
add("lzx21-r64", c, orders_for(len(c["files"]), 3))
# 3. multi-block frames and uncompressed blocks inside the intervals (block boundaries off the frame grid)
c = base(8, seed=13, block_mode=4, block_size=20000)
add("lzx21-r2-blocks", c, orders_for(len(c["files"]), 4))
# 4. E8 translation (x86-like plaintext, intel_filesize != 0): the origin is where the decoder was initialised
c = base(10, seed=15, text=2, intel_filesize=300000)
n = len(c["files"])
add("lzx21-r2-e8", c, orders_for(n, 5, extra=([n - 1], [n // 2, n // 2 + 1, n // 2 + 2], [3, 2, 1, 0])))
c = base(6, wb=17, seed=17, text=2, intel_filesize=12345678)
add("lzx17-r2-e8", c, orders_for(len(c["files"]), 6))
# 5. a damaged interval: every request whose decoder crosses it fails, the others do not
for fr, delta in ((6, 40), (7, 300), (0, 30)):
    c = base(8, wb=16, seed=19, mutations=[["flip_content", fr, delta, 4]])
    n = len(c["files"])
    add("lzx16-r2-flip@f%d" % fr, c, orders_for(n, 7, extra=([0, 1, 2, 3, 4, 5, 4, 5, 6, 7, 6], list(range(n)) * 2)))
# 6. UncompLen not a multiple of the interval (the reference pads it, chmd.c:1153-1157)
c = base(6, seed=21, uncomp_len=6 * I64 - 12345)
c["files"] = [f for f in c["files"] if f[1] + f[2] <= c["uncomp_len"]] + [["/tail.bin", 6 * I64 - 20000, 7655]]
add("lzx21-r2-shortlen", c, orders_for(len(c["files"]), 8))
# 7. unusable reset tables -> one stream from offset 0 with SpanInfo's length
c = base(6, seed=23, mutations=[["rtable_u32", 0x20, 0x4000]])              # FrameLen != 0x8000
add("lzx21-r2-badframelen", c, orders_for(len(c["files"]), 9))
c = base(6, seed=23, mutations=[["rtable_u32", 0x08, 6]])                   # entry size neither 4 nor 8
add("lzx21-r2-badentsize", c, orders_for(len(c["files"]), 10))
c = base(6, seed=23, mutations=[["rtable_u32", 0x04, 5]])                   # NumEntries: later intervals missing
add("lzx21-r2-fewentries", c, orders_for(len(c["files"]), 11))
c = base(6, seed=23, text=2, intel_filesize=77777, mutations=[["rtable_u32", 0x04, 5]])
add("lzx21-r2-fewentries-e8", c, orders_for(len(c["files"]), 12))
c = base(6, seed=23, mutations=[["rtable_u32", 0x20, 0x4000], ["spaninfo", 0]])   # ... and no usable SpanInfo
add("lzx21-r2-nospan", c, orders_for(len(c["files"]), 13)[:2])
# 8. reset-table entries that point beyond the content / at the wrong place
c = base(8, seed=25, mutations=[["rtable_entry", 6, 1 << 40]])
add("lzx21-r2-entry-far", c, orders_for(len(c["files"]), 14))
c = base(8, seed=25, mutations=[["rtable_entry", 6, 17]])
add("lzx21-r2-entry-wrong", c, orders_for(len(c["files"]), 15))
# 9. truncated file: the last intervals' compressed bytes are missing
c = base(8, seed=27, mutations=[["cut", 30000]])
add("lzx21-r2-cut", c, orders_for(len(c["files"]), 16))
# 10. a file that claims more bytes than the section holds; one that starts beyond it
c = base(4, seed=29)
c["files"] = c["files"] + [["/zz-long.bin", 4 * I64 - 1000, 5000], ["/zz-past.bin", 4 * I64 + 10, 10]]
add("lzx21-r2-overlong", c, orders_for(len(c["files"]), 17))
# 11. control data variants
c = base(4, seed=31, mutations=[["control_u32", 0x10, 3]])                  # window of 3 frames: not a power of two
add("lzx21-r2-badwindow", c, orders_for(len(c["files"]), 18)[:1])
c = base(4, seed=31, mutations=[["control_u32", 0x08, 3]])                  # ControlData version 3
add("lzx21-r2-badversion", c, orders_for(len(c["files"]), 19)[:1])
# 13. (round 4, found by tools/fuzz_chm_cpu.py) an UncompLen that LIES: files behind the stated end.  The reference looks the
# file's reset-table entry up before it looks at the length (chmd.c:1146-1157): a decoder created beyond the padded length gets a
# negative output length (lzxd_init == NULL -> MSPACK_ERR_NOMEMORY), one created exactly AT it an output length of 0 = "not known"
# (a request there is cut to one byte and succeeds), and a live decoder answers offsets beyond the length with MSPACK_ERR_DECRUNCH
c = base(8, wb=17, seed=35, n_files=9, uncomp_len=3 * I64 - 9000)
c["files"] = c["files"] + [["/zz-at-end.bin", 3 * I64, 700], ["/zz-one-more.bin", 3 * I64 + 700, 50]]
n = len(c["files"])
add("lzx17-r2-lying-uncomplen", c, orders_for(n, 21, extra=([n - 1], [n - 2], [n - 2, n - 1], [0, n - 2, 0, n - 1])))
# 14. (round 4, fuzz) ControlData's reset interval disagrees with the stream's: 3 frames -- not a power of two, so that the
# reference's rounding of UncompLen (& -interval) is no multiple of it -- and 65536 frames (2^31 bytes: negative as the int the
# reference computes in, lzxd_init(reset_interval < 0) == NULL -> MSPACK_ERR_NOMEMORY)
c = base(4, wb=17, rf=1, seed=37, n_files=8, mutations=[["control_u32", 0x0C, 3]])
add("lzx17-r1-interval3", c, orders_for(len(c["files"]), 22))
c = base(4, wb=17, rf=1, seed=37, n_files=8, mutations=[["control_u32", 0x0C, 65536]])
add("lzx17-r1-interval-negative", c, orders_for(len(c["files"]), 23)[:2])
# 12. BASELINE config 3: 1024 reset intervals (64 MiB), 96 files
c = base(1024, seed=33, n_files=96)
add("config3-1024-intervals", c, orders_for(len(c["files"]), 20)[:3])

out = os.path.join(HERE, "chm_extract.json")
json.dump(cases, open(out, "w"), separators=(",", ":"))
print("wrote", len(cases), "cases,", os.path.getsize(out), "bytes")
