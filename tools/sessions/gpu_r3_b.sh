#!/bin/bash
# round 3, second GPU session: pipe without eager take-over + lane-parallel map kernel; per-ticket timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r3b; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
B="python bench.py --exp --no-cpu --no-extras --steps 20 --warmup 5"
for cfg in "pipe:" "nopipe:MSPACK_HIP_NO_PIPE=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for u in 4096 1024 8192 16384; do
    ( env $envs timeout 200 $B --units $u 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$name units $u: ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'bit_exact', d['config']['bit_exact'], 'adopted', d['config']['units_on_frame_parallel_path'])
except Exception as e: print('$name units $u: FAILED', e)
" ) >> $OUT/bench.txt 2>&1
  done
done
for u in 4096 1024; do
  echo "== trace, $u units" >> $OUT/trace.txt
  MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace.so timeout 200 python tools/pipe_trace.py $u $OUT/trace_$u.npy >> $OUT/trace.txt 2>&1
done
( timeout 300 python -m pytest tests/test_gpu_lzx_frames.py -x -q -m gpu 2>&1 | tail -5 ) > $OUT/pytest.log 2>&1
cat $OUT/bench.txt $OUT/trace.txt $OUT/pytest.log
