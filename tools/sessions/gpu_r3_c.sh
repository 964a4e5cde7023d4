#!/bin/bash
# round 3, third GPU session: pipe v2 (parse waves store literals + match records, commit task resolves, unit kernel resumes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r3c; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
B="python bench.py --exp --no-cpu --no-extras --steps 20 --warmup 5"
for u in 4096 1024 8192 16384; do
  ( timeout 200 $B --units $u 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('pipe2 units $u: ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'bit_exact', d['config']['bit_exact'], 'adopted', d['config']['units_on_frame_parallel_path'])
except Exception as e: print('pipe2 units $u: FAILED', e)
" ) >> $OUT/bench.txt 2>&1
done
echo "== trace, 4096 units" >> $OUT/trace.txt
MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace.so timeout 200 python tools/pipe_trace.py 4096 $OUT/trace_4096.npy >> $OUT/trace.txt 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py --exp --no-cpu --no-extras --steps 10 --warmup 3 > $OUT/trace.log 2>&1 )
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do head -6 $f | cut -c1-60,250-400 > $OUT/kernel_stats_head.csv; done
( timeout 500 python -m pytest tests/test_gpu_lzx_frames.py tests/test_gpu_lzx.py tests/test_gpu_kat.py -x -q -m gpu 2>&1 | tail -8 ) > $OUT/pytest.log 2>&1
cat $OUT/bench.txt $OUT/trace.txt $OUT/kernel_stats_head.csv $OUT/pytest.log
