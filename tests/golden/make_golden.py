#!/usr/bin/env python3
"""Regenerates tests/golden/kat_folders.json -- run ONLY in the development container (needs
/root/reference and oracle/_ref).

Each entry is DATA taken from the reference's own test fixtures: the concatenated CFDATA payloads
of one folder (what cabd_sys_read feeds the codec; Quantum with the 0xFF trailers), its codec
parameters, and what the REAL reference codec (oracle/_ref, built from /root/reference) returns for
it: error code, bytes written, MD5 of the output.  Sources (all <= 40 KB):
  libmspack/test/test_files/cabd/*.cab   known answers + must-fail vectors of cabd_test.c:405-520
  cabextract/test/cabs/{mixed,large-files-cab}.cab, cabextract/test/bugs/*.cab
"""
import base64
import glob
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from helpers import cab_folders, folder_stream, ref_lzx, ref_mszip, ref_qtm  # noqa: E402

REF = "/root/reference"
FILES = sorted(glob.glob(REF + "/libmspack/test/test_files/cabd/*.cab")) + \
    sorted(glob.glob(REF + "/cabextract/test/bugs/*.cab")) + \
    [REF + "/cabextract/test/cabs/mixed.cab", REF + "/cabextract/test/cabs/large-files-cab.cab",
     REF + "/cabextract/test/cabs/simple.cab"]
KEEP = ("cve-", "lzx-", "mszip_lzx_qtm", "normal_2files", "qtm-", "mixed", "large-files-cab", "simple")

out = []
seen = set()
for path in FILES:
    name = os.path.basename(path)
    if not name.startswith(KEEP):
        continue
    cab = open(path, "rb").read()
    try:
        folders = cab_folders(cab)
    except Exception:
        continue
    for fi, f in enumerate(folders):
        if not f["blocks"]:
            continue
        ct = f["comp_type"]; method = ct & 15; wb = (ct >> 8) & 0x1F
        stream = folder_stream(f)
        total = sum(cu for _, cu in f["blocks"])
        key = hashlib.md5(stream + bytes([method, wb])).hexdigest()
        if key in seen or total == 0:
            continue
        # the reference's uninitialised-window reads make one vector non-deterministic: skip outputs
        if method == 1:
            err, data, w = ref_mszip(stream, total)
        elif method == 2 and 10 <= wb <= 21:
            err, data, w = ref_qtm(stream, total, wb)
        elif method == 3 and 15 <= wb <= 21:
            err, data, w = ref_lzx(stream, total, wb, 0, total)
        else:
            continue
        seen.add(key)
        out.append(dict(source=path[len(REF) + 1:], folder=fi, comp_type=ct, method=method, window_bits=wb,
                        out_len=total, n_blocks=len(f["blocks"]),
                        stream_b64=base64.b64encode(stream).decode(),
                        ref_err=err, ref_written=w, ref_md5=hashlib.md5(data).hexdigest(),
                        deterministic=("cve-2014-9556" not in name)))
json.dump(out, open(os.path.join(HERE, "kat_folders.json"), "w"), indent=1)
print("wrote", len(out), "vectors,", sum(len(o["stream_b64"]) for o in out), "b64 bytes")
