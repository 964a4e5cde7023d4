#!/bin/bash
# round 6, final session (GPU box): smoke, the whole -m gpu suite, the default bench line, rocprofv3 kernel stats of the bench command,
# the traffic passes, SQ counters, the trace build's phases -- what profiles/round6_final_* is made of
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${OUTDIR:-r6final}; mkdir -p $O
cd $R
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" ) > $O/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $O/summary.txt
timeout 2400 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=12 > $O/pytest_gpu.txt 2>&1; echo "pytest gpu rc=$?" | tee -a $O/summary.txt; tail -n 3 $O/pytest_gpu.txt >> $O/summary.txt
cp tests/_build/abort_bt.log $O/ 2>/dev/null
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $R/bench.py --no-cpu --no-extras --steps 20 --warmup 3 > $O/trace.log 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv 2>/dev/null
tail -n 1 $O/trace.log > $O/bench_under_rocprof.json
cd $R
rm -rf gpurun_out/traffic; bash tools/gpu_traffic.sh > $O/traffic.log 2>&1; cp gpurun_out/traffic/traffic.json $O/traffic_pmc.json 2>/dev/null
TAG=r6final bash tools/gpu_pmc_pipe.sh > $O/sq_counters.txt 2>&1
for n in 4096 8192; do
  echo "== trace build, ONE launch of $n intervals (MSPACK_HIP_NCHUNKS=1)" >> $O/phases.txt
  MSPACK_HIP_NCHUNKS=1 MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace.so timeout 300 python tools/pipe_trace.py $n >> $O/phases.txt 2>&1
done
bash tools/pmc_qtm.sh > $O/qtm_counters.txt 2>&1
cat $O/summary.txt; tail -n 1 $O/bench.json | cut -c1-600
