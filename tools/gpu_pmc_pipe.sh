#!/bin/bash
# SQ counters of the headline launch's kernels (separate --pmc passes, kernel-trace only)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_${TAG:-x}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
A="python $R/bench.py --exp --no-cpu --no-extras --steps 3 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/p1 -o pmc -- $A > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/p2 -o pmc -- $A > $OUT/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/p3 -o pmc -- $A > $OUT/p3.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("p1", "p2", "p3"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(float); n = collections.defaultdict(int)
        for row in csv.DictReader(open(f)):
            k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            agg[k] += float(row["Counter_Value"]); n[k] += 1
        for k in sorted(agg):
            if "mspack" in k[0]: print("%-22s %-24s %.6g  (dispatches %d)" % (k[0], k[1], agg[k] / n[k], n[k]))
PY
