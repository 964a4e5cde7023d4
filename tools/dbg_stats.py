# scratch: section timers of the LZX speculative path (build with -DLZX_EXP_STATS, run with MSPACK_HIP_SO=...)
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import libmspack_amd as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ub = 65536
plain, comp, off, ln = M.corpus_lzx_units(0xBA5E11, 0, n, ub, 21)
units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(n, ub), window_bits=21, reset_frames=2)
out, res = M.decode_batch(units, comp, out_bytes)
oo = np.asarray(units['out_off'], dtype=np.int64)
raw = np.stack([np.frombuffer(out[o:o + 64].tobytes(), dtype=np.uint32) for o in oo]).astype(np.int64)
tot = raw[:, 6].mean(); rounds = raw[:, 7].mean()
print("units", n, "total ticks/unit %.0f  rounds/unit %.1f  ticks/round %.0f  spec-run ticks (>>6) %.0f" % (tot, rounds, tot / rounds, raw[:, 8].mean()))
for name, c in [("parse: vector decode", 0), ("parse: chain walk", 1), ("parse: queue tokens + scalar tokens", 2),
                ("commit: read, scan, literals", 3), ("commit: R0-R2", 4), ("commit: checks + queue", 5), ("flush (queue resolve)", 12)]:
    v = raw[:, c].mean(); print("  %-36s %10.0f  %5.1f%%  %7.0f /round" % (name, v, 100 * v / tot, v / rounds))
print("  %-24s %10.0f  %5.1f%%" % ("outside the round loop", tot - raw[:, :6].sum(1).mean() - raw[:, 12].mean(), 100 * (tot - raw[:, :6].sum(1).mean() - raw[:, 12].mean()) / tot))
for name, c in [("scalar token loop", 15)]:
    print("    %-24s %10.0f  %5.1f%%" % (name, raw[:, c].mean(), 100 * raw[:, c].mean() / tot))
print("  headers: total %.0f  pretree read+build %.0f  length symbols %.0f  main/len table builds %.0f" % (raw[:, 8].mean() * 64, raw[:, 9].mean(), raw[:, 10].mean(), raw[:, 11].mean()))
t = raw[:, 6].astype(float)
print("unit time (ticks): min %.0f  p10 %.0f  median %.0f  p90 %.0f  p99 %.0f  max %.0f  mean %.0f" %
      (t.min(), np.percentile(t, 10), np.median(t), np.percentile(t, 90), np.percentile(t, 99), t.max(), t.mean()))
r = np.asarray(ln, dtype=float) / ub
for lo, hi in [(0, .2), (.2, .3), (.3, .4), (.4, .5), (.5, .7), (.7, 2)]:
    m = (r >= lo) & (r < hi)
    if m.any(): print("  ratio %.1f-%.1f: %4d units, mean time %.0f, rounds %.0f" % (lo, hi, m.sum(), t[m].mean(), raw[m, 7].mean()))
