"""Round 5, VERDICT item 1: look for the intermittent dirty interval of the config-3 CHM batch.

  python tools/stress_chm_batch.py SECONDS [mix]

Decodes the 1024-interval batch the way chmd.c:decode_intervals builds it (every unit's input runs to the end of the arena)
over and over for SECONDS; with `mix`, batches of other shapes (a few intervals of another CHM, MSZIP units) are decoded in
between so that the persistent context's buffers are reused at other sizes.  Every anomaly is printed with the unit's result,
the result of decoding that unit ALONE right afterwards, and the first differing byte."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libmspack_amd as M                      # noqa: E402

VECS = json.load(open(os.path.join(ROOT, "tests", "golden", "chm_extract.json")))
V = [v for v in VECS if v["tag"].startswith("config3")][0]


def chm_batch(case, n_bytes=None):
    n = n_bytes or case["n_bytes"]
    d = M.gen_plaintext(case["seed"], case["text"], n)
    o = M.lzx_opts(mode=case.get("block_mode", 0), block_size=case.get("block_size", 0),
                   intel_filesize=case.get("intel_filesize", 0), e8_base=0)
    lz, fo = M.lzx_encode(d, case["window_bits"], case["reset_frames"], o)
    fper = case["reset_frames"]
    nint = (len(fo) - 1) // fper
    lz = np.asarray(lz, dtype=np.uint8)
    base = (lz.size + 128 + 3) & ~3
    arena = np.zeros(base + 4 * nint * fper + 64, dtype=np.uint8)
    arena[:lz.size] = lz
    tab = arena[base:base + 4 * nint * fper].view("<u4")
    offs = np.asarray(fo[:nint * fper:fper], dtype=np.int64)
    fo = np.asarray(fo, dtype=np.int64)
    tab[:] = (fo[:nint * fper] - np.repeat(offs, fper)).astype(np.uint32)

    def units():
        return M.make_units(M.KIND_LZX, offs, lz.size - offs, [fper * 32768] * nint, window_bits=case["window_bits"],
                            reset_frames=fper, e8_base=[k * fper * 32768 for k in range(nint)],
                            frame_tabs=[base + 4 * k * fper for k in range(nint)])
    return d, arena, units, nint, fper


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    mix = len(sys.argv) > 2
    d, arena, units, nint, fper = chm_batch(V["case"])
    small = chm_batch(dict(V["case"], seed=5), n_bytes=40 * 65536) if mix else None
    t_end = time.time() + secs
    it = bad = 0
    rng = np.random.default_rng(1)
    while time.time() < t_end:
        it += 1
        if mix and it % 3 == 0:
            sd, sa, su, sn, _ = small
            u, ob = su()
            k = int(rng.integers(1, sn))
            out, res = M.decode_batch(u[:k], sa, ob + 64)
            if not np.array_equal(out[:k * 65536], sd[:k * 65536]) or (res["err"][:k - 1] != 0).any():
                print("iter %d: small batch (%d units) wrong" % (it, k), flush=True)
                bad += 1
        u, out_bytes = units()
        try:
            out, res = M.decode_batch(u, arena, out_bytes + 64)
        except M.MspackHipError as e:
            print("iter %d: FAILED %s" % (it, e), flush=True)
            bad += 1
            continue
        clean = (res["err"] == 0) | ((res["err"] == M.ERR_READ) & ((res["flags"] & M.F_LOOKAHEAD_READ) != 0))
        same = np.array_equal(out[:d.size], d)
        if clean.all() and same:
            continue
        bad += 1
        dirty = np.nonzero(~clean)[0]
        diff = np.nonzero(out[:d.size] != d)[0]
        print("iter %d: dirty units %s; first differing byte %s (unit %s)" % (
            it, [int(k) for k in dirty[:8]], int(diff[0]) if diff.size else None,
            int(diff[0]) // (fper * 32768) if diff.size else None), flush=True)
        for k in list(dirty[:4]) + ([int(diff[0]) // (fper * 32768)] if diff.size else []):
            k = int(k)
            print("   unit %d batch result %s" % (k, res[k]), flush=True)
            u1, ob1 = units()
            o1, r1 = M.decode_batch(u1[k:k + 1].copy(), arena, ob1 + 64)
            print("   unit %d alone        %s bytes %s" % (k, r1[0], "ok" if np.array_equal(
                o1[k * fper * 32768:(k + 1) * fper * 32768], d[k * fper * 32768:(k + 1) * fper * 32768]) else "MISMATCH"), flush=True)
    print("stress: %d iterations, %d anomalies" % (it, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
