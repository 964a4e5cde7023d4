"""scratch: what the cabinet driver SAYS (sys->message) against what the REAL reference says, on the cabinets of tools/fuzz_drivers_cpu.py
(four folders, random damage; the drivers on the CPU stand-in for the batch ABI): the lines of open() and of every extract() call, in
order, plain and salvage mode.  This side's callback sees the format strings only (a ctypes callback cannot read C varargs): a line
matches when the reference's formatted line fits the format.  MSZIP repair mode is not run here (tests/test_gpu_messages.py does).
    python tools/fuzz_cab_messages_cpu.py <seed> [cases]"""
import os, re, sys
import numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_); sys.path.insert(0, os.path.join(R_, "tests")); sys.path.insert(0, os.path.join(R_, "tools"))
from libmspack_amd import api
import helpers
import fuzz_drivers_cpu as F


def fits(fmt, line):
    pat = re.escape(fmt.decode("latin-1"))
    pat = re.sub(r"%(?:l|ll)?[udx]", r"-?[0-9a-fA-F]+", pat.replace("\\%", "%"))
    pat = pat.replace("%s", ".*")
    return re.fullmatch(pat, line, re.S) is not None


def same(mine, ref):
    return len(mine) == len(ref) and all(fits(m, r) for m, r in zip(mine, ref))


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    assert helpers.have_ref()
    helpers.ref().refh_zero_alloc(1)
    L = F.hostlogic()
    rng = np.random.default_rng(seed)
    bad = 0
    for k in range(cases):
        if k % 25 == 0: cab0 = F.base_cab(seed * 1000 + k)
        cab = F.mutate(cab0, rng) if k % 10 else bytes(cab0)
        for salvage in (0, 1):
            for order in ([0, 1, 2, 3, 4, 5, 6, 7], [7, 6, 5, 4, 3, 2, 1, 0], [1, 1, 0, 5, 4, 5]):
                helpers.ref_messages()
                rc, want = helpers.ref_cab_extract(cab, order, cap=len(order) * 160000 + 4096, salvage=salvage)
                lines = helpers.ref_messages()
                ref_open, ref_calls, ref_h_open, ref_h_calls = [], [], [], []      # (lines, and per line: said with a file handle?)
                for l, h in zip(lines, helpers.ref_message_handles()):
                    if l.startswith("#extract"): ref_calls.append([]); ref_h_calls.append([])
                    elif ref_calls: ref_calls[-1].append(l); ref_h_calls[-1].append(h == "H")
                    else: ref_open.append(l); ref_h_open.append(h == "H")
                with api.Cab(cab, mem=True, L=L, salvage=salvage) as c:
                    if c.open_error:
                        if rc == 0: bad += 1; print("case %d: open mine %d, the reference opens it" % (k, c.open_error))
                        break
                    if rc: break
                    if not same(c.mem.messages, ref_open) or c.mem.message_handles != ref_h_open:
                        bad += 1; print("case %d salvage %d open: reference %s mine %s" % (k, salvage, ref_open, c.mem.messages)); break
                    ok = True
                    for j, i in enumerate(order):
                        if i >= len(c.files): break
                        del c.mem.messages[:]
                        del c.mem.message_handles[:]
                        c.mem.outputs.clear()
                        err, data = c.extract(i)
                        if j < len(ref_calls) and (not same(c.mem.messages, ref_calls[j]) or c.mem.message_handles != ref_h_calls[j]):
                            bad += 1; ok = False
                            print("case %d salvage %d order %s call %d (file %d, err %d / %d): reference %s mine %s" %
                                  (k, salvage, order, j, i, want[j][0], err, list(zip(ref_calls[j], ref_h_calls[j])), list(zip(c.mem.messages, c.mem.message_handles)))); break
                    if not ok: break
            else: continue
            break
    print("seed %d: %d cases, %d mismatches" % (seed, cases, bad))


if __name__ == "__main__":
    main()
