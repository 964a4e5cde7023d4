"""The LZX codec's warning through sys->message (missing #4): "WARNING; invalid reset interval detected during
LZX decompression" -- a block that is still open when the decoder reaches a reset point (lzxd.c:423-431), said once per
lzxd_decompress call that meets such a reset, after which decoding starts over at the reset point and goes on.
tests/golden/chm_messages.json holds what the REAL libmspack answered and said for four recipe CHMs (a block header's length
field raised by a few bytes; tests/golden/make_chm_messages_golden.py) in several extraction orders; the same calls go through
include/mspack.h here and every extract() call must return the same code and bytes and say the same lines.
  * `-m gpu`: libmspack_hip.so -- the kernels report the reset points in the unit's log (MSPACK_HIP_UF_LZX_LOG);
  * `-m "not gpu"`: the same driver code (csrc/host/chmd.c) on the CPU stand-in for the batch ABI (host logic only)."""
import hashlib
import json
import os

import pytest

from libmspack_amd import api
import chm_extract_recipe as R

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "chm_messages.json")))


def per_call(lines):
    calls = []
    for l in lines:
        if l.startswith("#extract"):
            calls.append([])
        else:
            calls[-1].append(l)
    return calls


def replay(v, L=None):
    chm, _d, files = R.build(v["case"])
    assert hashlib.md5(chm).hexdigest() == v["chm_md5"], "recipe no longer reproduces the golden CHM"
    for run in v["runs"]:
        want = per_call(run["messages"])
        with api.Chm(chm, mem=True, L=L) as c:
            assert c.open_error == 0
            for k, (idx, exp) in enumerate(zip(run["order"], run["results"])):
                before = len(c.mem.messages)
                err, data = c.extract(idx)
                said = [m.decode("latin1") if isinstance(m, bytes) else m for m in c.mem.messages[before:]]
                tag = "%s order %s call %d (file %d)" % (v["tag"], run["order"], k, idx)
                assert err == exp["err"] and len(data) == exp["n"] and hashlib.md5(data).hexdigest() == exp["md5"], tag
                assert said == want[k], (tag, said, want[k])


def test_golden_has_warnings():
    n = sum(l.startswith("WARNING; invalid reset interval") for v in GOLD for r in v["runs"] for l in r["messages"])
    assert n >= 10


@pytest.mark.gpu
@pytest.mark.parametrize("v", GOLD, ids=[v["tag"] for v in GOLD])
def test_lzx_reset_warning_gpu(built, v):
    replay(v)


@pytest.mark.parametrize("v", GOLD, ids=[v["tag"] for v in GOLD])
def test_lzx_reset_warning_host_logic_cpu(built, hostlogic, v):
    replay(v, L=hostlogic)
