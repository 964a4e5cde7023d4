"""GPU parity on the reference's own known-answer and must-fail vectors (tests/golden/kat_folders.json,
generated from libmspack/test/test_files/cabd and cabextract/test by tests/golden/make_golden.py):
error code, byte count and MD5 must equal what the real reference codec returned."""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

import libmspack_amd as M

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kat_folders.json")))


def test_kat_folders_gpu(built):
    streams = [base64.b64decode(k["stream_b64"]) for k in KAT]
    offs, pos = [], 0
    for s in streams:
        pos = (pos + 15) & ~15
        offs.append(pos); pos += len(s)
    arena = np.zeros(pos + 64, dtype=np.uint8)
    for s, o in zip(streams, offs):
        arena[o:o + len(s)] = np.frombuffer(s, dtype=np.uint8)
    units, out_bytes = M.make_units([k["method"] for k in KAT], offs, [len(s) for s in streams],
                                    [k["out_len"] for k in KAT], window_bits=[k["window_bits"] for k in KAT],
                                    out_slack=32768)
    out, res = M.decode_batch(units, arena, out_bytes)
    for i, k in enumerate(KAT):
        tag = "%s folder %d" % (k["source"], k["folder"])
        assert res["err"][i] == k["ref_err"], (tag, res[i])
        assert res["out_len"][i] == k["ref_written"], (tag, res[i])
        if k["deterministic"]:
            got = out[units["out_off"][i]:units["out_off"][i] + k["ref_written"]].tobytes()
            assert hashlib.md5(got).hexdigest() == k["ref_md5"], tag
