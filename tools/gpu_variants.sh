#!/bin/bash
# bench the headline launch (and 8192 units) with analysis / tuning builds from build/variants:  VARIANTS="a b c" bash tools/gpu_variants.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/variants; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
B="python bench.py --exp --no-cpu --no-extras --steps 15 --warmup 4"
for rep in 1 ${REPS:+2}; do for v in base $VARIANTS; do
  so=$R/build/variants/libmspack_hip_$v.so; [ $v = base ] && so=$R/libmspack_amd/libmspack_hip.so
  for u in ${UNITS:-4096 8192}; do
    ( MSPACK_HIP_SO=$so timeout 200 $B --units $u 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('%-10s units %5d: ms_per_step %7.3f bit_exact %s adopted %s' % ('$v', $u, d['ms_per_step'], d['config']['bit_exact'], d['config']['units_on_frame_parallel_path']))
except Exception as e: print('$v units $u: FAILED', e)
" ) >> $OUT/bench_${TAG:-x}.txt 2>&1
  done
done; done
if [ -n "$TRACE" ]; then
  echo "== trace, 4096 units" > $OUT/trace_${TAG:-x}.txt
  MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace.so timeout 200 python tools/pipe_trace.py 4096 >> $OUT/trace_${TAG:-x}.txt 2>&1
  cat $OUT/trace_${TAG:-x}.txt
fi
cat $OUT/bench_${TAG:-x}.txt
