# scratch: static instruction counts between "; MARK" comments of an -S dump built with -DLZX_MARKS
import re, sys
lines = open(sys.argv[1]).read().splitlines()
marks = [(i, l.split("MARK")[1].strip()) for i, l in enumerate(lines) if "; MARK" in l]
def count(a, b):
    v = s = ds = vm = br = other = 0
    for l in lines[a:b]:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
        op = t.split()[0]
        if op.startswith("v_"): v += 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch"): br += 1
        elif op.startswith("s_"): s += 1
        elif op.startswith("ds_"): ds += 1
        elif op.startswith(("global_", "flat_", "buffer_", "scratch_")): vm += 1
        else: other += 1
    return v, s, ds, vm, br, other
for (i, n), (j, m) in zip(marks, marks[1:]):
    c = count(i, j)
    print("%-18s -> %-18s lines %5d  VALU %4d SALU %4d DS %3d VMEM %3d BR %3d other %d" % ((n, m, j - i) + c))
