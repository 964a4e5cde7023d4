#!/bin/bash
# round 4: MSZIP parse waves with the balanced last walk + literal ring (A/B against the build before: variant zipold)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/zipring; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_mszip_blocks.py tests/test_gpu_mszip.py tests/test_config2_cab.py tests/test_gpu_messages.py -q -x -m gpu 2>&1 | tail -3 ) > $OUT/tests.log 2>&1
cat $OUT/tests.log
( VARIANTS=zipold timeout 600 bash tools/gpu_mszip_ab.sh 2>&1 | grep -v "^$" ) > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
cd /tmp
for c in WRITE_SIZE; do
  ( cd $R && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/mz_$c -o pmc -- python tools/bench_mszip_folder.py 4096 1 > $OUT/mz_$c.log 2>&1 )
done
python - <<PY
import csv, glob, collections
for c in ("WRITE_SIZE",):
    per = collections.defaultdict(list)
    for f in glob.glob("$OUT/mz_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if "mszip" in k: per[k].append(float(row["Counter_Value"]))
    for k in sorted(per):
        v = per[k]
        print("%-12s %-24s dispatches %3d  KiB per dispatch: first %.0f  last %.0f  mean %.0f" % (c, k, len(v), v[0], v[-1], sum(v) / len(v)))
PY
