#!/bin/bash
# round 3, first GPU session: the dependency-driven LZX launch (mspack_lzx_pipe) against the three-kernel path and the
# serial kernel; parity tests of the LZX paths; kernel trace.     gpurun --timeout 900 -- 'bash tools/gpu_r3_a.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r3a; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_gpu_lzx_frames.py tests/test_gpu_lzx.py tests/test_gpu_kat.py tests/test_gpu_hostpath.py -x -q -m gpu 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1
B="python bench.py --exp --no-cpu --no-extras --steps 20 --warmup 5"
for cfg in "pipe:" "nopipe:MSPACK_HIP_NO_PIPE=1" "pipe12:MSPACK_HIP_PIPE_WAVES_PER_CU=12" "pipe20:MSPACK_HIP_PIPE_WAVES_PER_CU=20"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  for u in 4096 1024 8192; do
    ( env $envs timeout 200 $B --units $u 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('$name units $u: ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'bit_exact', d['config']['bit_exact'], 'adopted', d['config']['units_on_frame_parallel_path'])
except Exception as e: print('$name units $u: FAILED', e)
" ) >> $OUT/bench.txt 2>&1
  done
done
( timeout 200 $B --units 4096 --no-frame-tables 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('serial units 4096: ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'bit_exact', d['config']['bit_exact'])" ) >> $OUT/bench.txt 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --exp --no-cpu --no-extras --steps 10 --warmup 3 > $OUT/trace.log 2>&1
cd $R; for f in $(find $OUT/trace -name "*kernel_stats.csv"); do head -8 $f > $OUT/kernel_stats_head.csv; done
cat $OUT/pytest.log $OUT/bench.txt $OUT/kernel_stats_head.csv
