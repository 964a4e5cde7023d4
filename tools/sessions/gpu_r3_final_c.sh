#!/bin/bash
# round 3, last GPU session: smoke(), the whole -m gpu suite, the default bench line (what the driver runs), rocprofv3 kernel stats of the
# bench command, FETCH_SIZE / WRITE_SIZE of MSZIP config 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r3g; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -3 ) > $OUT/smoke.log 2>&1
( timeout 1500 python -m pytest tests -x -q -m gpu --durations=6 2>&1 | tail -14 ) > $OUT/pytest.log 2>&1
( timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --no-cpu --no-extras --steps 20 --warmup 5 > $OUT/stats_bench.json 2> $OUT/stats.err )
for f in $(find $OUT/stats -name "*kernel_stats.csv"); do cp $f $OUT/bench_kernel_stats.csv; done
for c in FETCH_SIZE WRITE_SIZE; do
  ( timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/mz_$c -o pmc -- python tools/bench_mszip_folder.py 4096 1 > $OUT/mz_$c.log 2>&1 )
done
python - <<PY > $OUT/traffic_mszip.txt 2>&1
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(list)
    for f in glob.glob("$OUT/mz_%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if "mszip" in k or "frame_map" in k: per[k].append(float(row["Counter_Value"]))
    for k in sorted(per):
        v = per[k]
        print("%-12s %-24s dispatches %3d  KiB per dispatch: %s" % (c, k, len(v), " ".join("%.0f" % x for x in v)))
PY
cat $OUT/smoke.log $OUT/pytest.log; tail -c 600 $OUT/bench.json; echo; head -8 $OUT/bench_kernel_stats.csv | cut -c1-60,200-330; cat $OUT/traffic_mszip.txt
