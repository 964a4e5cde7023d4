#!/bin/bash
# round 3, final validation B: rocprofv3 kernel stats of the bench command, HBM traffic (LZX headline, MSZIP config 2),
# SQ counters of the pipe kernel, per-phase sums of the trace build
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r3f; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --no-cpu --steps 20 --warmup 5 > $OUT/stats_bench.json 2> $OUT/stats.err )
for f in $(find $OUT/stats -name "*kernel_stats.csv"); do cp $f $OUT/bench_kernel_stats.csv; done
( cd $R && timeout 400 bash tools/gpu_traffic.sh > $OUT/traffic_lzx.txt 2>&1; cp gpurun_out/traffic/traffic.json $OUT/traffic_lzx.json )
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd $R && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/mz_$c -o pmc -- python tools/bench_mszip_folder.py 4096 1 > $OUT/mz_$c.log 2>&1 )
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
        print("%-12s %-24s dispatches %3d  KiB per dispatch: first %.0f  last %.0f  mean %.0f" % (c, k, len(v), v[0], v[-1], sum(v) / len(v)))
PY
( cd $R && TAG=final timeout 400 bash tools/gpu_pmc_pipe.sh > $OUT/sq_counters.txt 2>&1 )
( cd $R && MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace.so timeout 200 python tools/pipe_trace.py 4096 > $OUT/phases.txt 2>&1 )
head -12 $OUT/bench_kernel_stats.csv | cut -c1-70,200-330; cat $OUT/traffic_lzx.json | head -8; cat $OUT/traffic_mszip.txt; tail -30 $OUT/phases.txt
