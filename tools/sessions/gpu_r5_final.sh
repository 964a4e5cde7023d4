#!/bin/bash
# round 5, final validation: smoke, the whole -m gpu suite as the driver runs it, the default bench line; then the evidence behind it:
# rocprofv3 kernel stats of the bench command, HBM-side traffic (separate --pmc passes), per-phase sums of the trace build, Quantum counters
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5final; mkdir -p $OUT; cd $R
python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -10 $OUT/pytest_gpu.log
( timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cd /tmp; export TMPDIR=/tmp
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python $R/bench.py --no-cpu --no-extras --steps 20 --warmup 3 > $OUT/stats_bench.json 2> $OUT/stats.err )
for f in $(find $OUT/stats -name "*kernel_stats.csv"); do cp $f $OUT/bench_kernel_stats.csv; done
( cd $R && timeout 300 bash tools/gpu_traffic.sh > $OUT/traffic_lzx.txt 2>&1; cp gpurun_out/traffic/traffic.json $OUT/traffic_lzx.json )
( cd $R && MSPACK_HIP_NCHUNKS=1 MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace.so timeout 200 python tools/pipe_trace.py 4096 > $OUT/phases.txt 2>&1 )
( cd $R && timeout 400 bash tools/pmc_qtm.sh > $OUT/qtm_counters.txt 2>&1 )
cd $R
head -8 $OUT/bench_kernel_stats.csv | cut -c1-60,200-330; cat $OUT/traffic_lzx.json | head -8; tail -16 $OUT/phases.txt; tail -14 $OUT/qtm_counters.txt
python - <<'P'
import json
d = json.loads(open("gpurun_out/r5final/bench.json").read().strip().splitlines()[-1])
print({k: d.get(k) for k in ("value", "ms_per_step", "value_host_inclusive", "value_host_to_host")}, d["roofline"]["frac"])
for s in d.get("secondary", []):
    print(s.get("config", "")[:50], s.get("kernel_ms"), (s.get("through_api") or {}).get("MBps"))
P
