"""cabextract's large-files.test (cabextract/test/large-files.test:13-24): `large-files-cab.cab` holds `large-files.cab`, whose
three folders -- MSZIP, LZX window 2^15, LZX window 2^21 -- are 65 535 CFDATA blocks = 2 147 450 880 bytes each, made by
Microsoft's encoder; all three extract to the MD5 d64bf04a56027b97ac17d751aba2d291.  They are the reference's largest
known-answer vectors and sit at every 32-bit edge of this code (unit lengths, positions, frame slots, scratch sizes).

The inner cabinet is not a fixture: it is DECODED here from the outer folder's compressed stream, which
tests/golden/kat_folders.json holds (14.7 MB, MD5 pinned there), then opened through the libmspack object API.
Slow (a folder is one unit: its 65 535 frames commit one after the other) and large (6 GiB of output, ~26 GiB of work
scratch on the device): opt in with MSPACK_TEST_LARGE=1.  `python tests/test_gpu_large_files.py` prints the timings."""
import base64
import hashlib
import json
import os
import sys
import tempfile
import time

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # (also runs as a script)
import libmspack_amd as M
from libmspack_amd import api

MD5 = "d64bf04a56027b97ac17d751aba2d291"
SIZE = 2147450880
KAT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_folders.json")


def inner_cabinet():
    k = [v for v in json.load(open(KAT)) if v["source"].endswith("large-files-cab.cab")][0]
    s = base64.b64decode(k["stream_b64"])
    arena = np.zeros(len(s) + 64, dtype=np.uint8); arena[:len(s)] = np.frombuffer(s, dtype=np.uint8)
    units, out_bytes = M.make_units([k["method"]], [0], [len(s)], [k["out_len"]], window_bits=[k["window_bits"]], out_slack=32768)
    out, res = M.decode_batch(units, arena, out_bytes)
    assert res["err"][0] == 0 and res["out_len"][0] == k["out_len"]
    cab = out[:k["out_len"]].tobytes()
    assert hashlib.md5(cab).hexdigest() == k["ref_md5"]
    return cab


def run(report=print):
    cab = inner_cabinet()
    tmp = tempfile.mkdtemp(prefix="mspk_large_")
    path = os.path.join(tmp, "large-files.cab")
    open(path, "wb").write(cab)
    results = []
    with api.Cab(path) as c:
        assert c.open_error == 0
        names = [f[0] for f in c.files]
        assert names == [b"mszip-2gb.txt", b"lzx15-2gb.txt", b"lzx21-2gb.txt"], names
        assert all(f[1] == SIZE for f in c.files)
        for i, nm in enumerate(names):
            out = os.path.join(tmp, "out%d" % i)
            t0 = time.time()
            err = c.d.contents.extract(c.d, c._files[i], os.fsencode(out))
            dt = time.time() - t0
            h = hashlib.md5()
            with open(out, "rb") as fh:
                while True:
                    b = fh.read(1 << 24)
                    if not b:
                        break
                    h.update(b)
            n = os.path.getsize(out)
            os.unlink(out)
            report("%-14s err %d  %d bytes  md5 %s  extract() %.1f s" % (nm.decode(), err, n, h.hexdigest(), dt))
            results.append((err, n, h.hexdigest()))
    os.unlink(path); os.rmdir(tmp)
    return results


@pytest.mark.gpu
def test_large_files_prefix(built):
    """the same cabinet with every folder cut to its first 4096 CFDATA blocks (128 MiB each; tests/helpers.cab_cut_folders):
    runs by default, so that the three Microsoft-made folders -- MSZIP, LZX-15 and LZX-21 units of 4096 frames, each one
    long chain of frames -- are covered without the opt-in.  Expected MD5s: what the real libmspack extracts from the same
    cut cabinet (tests/golden/large_prefix.json, made by tests/golden/make_large_prefix_golden.py with oracle/_ref)."""
    from helpers import cab_cut_folders
    gold = json.load(open(os.path.join(os.path.dirname(KAT), "large_prefix.json")))
    cab = cab_cut_folders(inner_cabinet(), gold["n_blocks"])
    with api.Cab(cab, mem=True) as c:
        assert c.open_error == 0 and len(c.files) == 3
        for g in gold["files"]:
            t0 = time.time()
            err, out = c.extract(g["index"])
            dt = time.time() - t0
            assert err == 0 and len(out) == g["bytes"] == gold["length"], (g["index"], err, len(out))
            assert hashlib.md5(out).hexdigest() == g["md5"], g["index"]
            print("folder %d: %d bytes in %.2f s (%.0f MB/s through cabd->extract())" % (g["index"], len(out), dt, len(out) / dt / 1e6))


if os.environ.get("MSPACK_TEST_LARGE"):      # (opt-in: slow and large; not collected otherwise -- the prefix test above always runs)
    @pytest.mark.gpu
    def test_large_files(built):
        for err, n, md5 in run():
            assert err == 0 and n == SIZE and md5 == MD5


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    t0 = time.time()
    r = run()
    print("all three match:", all(x == (0, SIZE, MD5) for x in r), "total %.1f s" % (time.time() - t0))
