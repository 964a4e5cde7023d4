"""Round 5, VERDICT item 1: many decompressor lifetimes in one process (the config-3 CHM of tests/golden/chm_extract.json).

  python tools/repro_chm_lifetimes.py driver [N]   N x (create -> open -> extract(last file) -> extract(first) -> close -> destroy)
  python tools/repro_chm_lifetimes.py units  [N]   the same 1024-interval batch, N times, through mspack_hip_decode_batch the way
                                                   chmd.c:decode_intervals builds it (every unit's input runs to the end of the arena)
Prints one line per lifetime / call: error code, sys->message lines, mspack_hip_last_error(), dirty units."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libmspack_amd as M                      # noqa: E402
from libmspack_amd import api                  # noqa: E402
import chm_extract_recipe as R                 # noqa: E402

VECS = json.load(open(os.path.join(ROOT, "tests", "golden", "chm_extract.json")))
V = [v for v in VECS if v["tag"].startswith("config3")][0]


def driver(n):
    chm, _d, files = R.build(V["case"])
    run = V["runs"][1]
    want = {idx: exp for idx, exp in zip(run["order"], run["results"])}
    bad = 0
    for it in range(n):
        with api.Chm(chm, mem=True) as c:
            line = []
            for idx in (len(files) - 1, 0, len(files) // 2):
                err, data = c.extract(idx)
                ok = err == want[idx]["err"] and hashlib.md5(data).hexdigest() == want[idx]["md5"]
                line.append("f%d err=%d %s" % (idx, err, "ok" if ok else "MISMATCH"))
                bad += 0 if ok else 1
            print("lifetime %d: %s | messages=%r | last_error=%r" % (
                it, "; ".join(line), c.mem.messages, M.lib().mspack_hip_last_error().decode()), flush=True)
    return bad


def units(n):
    case = V["case"]
    d = M.gen_plaintext(case["seed"], case["text"], case["n_bytes"])
    o = M.lzx_opts(mode=case.get("block_mode", 0), block_size=case.get("block_size", 0),
                   intel_filesize=case.get("intel_filesize", 0), e8_base=0)
    lz, fo = M.lzx_encode(d, case["window_bits"], case["reset_frames"], o)
    fper = case["reset_frames"]
    nint = (len(fo) - 1) // fper
    lz = np.frombuffer(lz, dtype=np.uint8) if not isinstance(lz, np.ndarray) else lz
    base = (lz.size + 128 + 3) & ~3
    arena = np.zeros(base + 4 * nint * fper + 64, dtype=np.uint8)
    arena[:lz.size] = lz
    tab = arena[base:base + 4 * nint * fper].view("<u4")
    offs = np.asarray(fo[:nint * fper:fper], dtype=np.int64)
    for k in range(nint):
        for j in range(fper):
            tab[k * fper + j] = int(fo[k * fper + j]) - int(offs[k])
    bad = 0
    for it in range(n):
        u, out_bytes = M.make_units(M.KIND_LZX, offs, lz.size - offs, [fper * 32768] * nint, window_bits=case["window_bits"],
                                    reset_frames=fper, e8_base=[k * fper * 32768 for k in range(nint)],
                                    frame_tabs=[base + 4 * k * fper for k in range(nint)])
        try:
            out, res = M.decode_batch(u, arena, out_bytes + 64)
        except M.MspackHipError as e:
            print("call %d: FAILED %s" % (it, e), flush=True)
            bad += 1
            continue
        clean = (res["err"] == 0) | ((res["err"] == M.ERR_READ) & ((res["flags"] & M.F_LOOKAHEAD_READ) != 0))
        same = np.array_equal(out[:d.size], d)
        dirty = np.nonzero(~clean)[0]
        print("call %d: dirty units %s bytes %s" % (it, [(int(k), int(res["err"][k]), hex(int(res["flags"][k])), int(res["good_len"][k]))
                                                        for k in dirty[:12]], "ok" if same else "MISMATCH"), flush=True)
        bad += len(dirty) + (0 if same else 1)
    return bad


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "driver"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    rc = driver(n) if what == "driver" else units(n)
    print("%s: %d problems" % (what, rc))
    sys.exit(1 if rc else 0)
