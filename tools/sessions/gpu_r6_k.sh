#!/bin/bash
# round 6, session K (GPU box): the lane parser's first walk with a shorter tail (LZX_LANE_TAIL 128 / 192 / 256 against 384: with a
# 3 KiB stage a stretch is 384 bits, so the first round walked whole stretches) -- headline and 8192 units, two repetitions; the
# launch-path parity test with the streaming resolve off
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6k; mkdir -p $O
cd $R
timeout 900 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_lzx_frames.py -k launch_paths > $O/parity.log 2>&1; echo "launch paths rc=$?" | tee -a $O/summary.txt; tail -3 $O/parity.log >> $O/summary.txt
VARIANTS="tail128 tail192 tail256" REPS=1 TAG=tail bash tools/gpu_variants.sh > $O/variants.txt 2>&1
cp gpurun_out/variants/bench_tail.txt $O/ 2>/dev/null
cat $O/summary.txt; cat $O/bench_tail.txt
