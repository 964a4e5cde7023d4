#!/bin/bash
# round 6, session W (GPU box): ticket order with parse and resolve tasks side by side (P(f0) | P(f1) and R(f0) alternating | R(f1)) against
# the level order -- headline and 8192 units, two repetitions; the trace build's phases; parity of the frame tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6w; mkdir -p $O
cd $R
VARIANTS="mixed" REPS=1 TAG=mixed UNITS="1024 4096 8192" bash tools/gpu_variants.sh > $O/variants.txt 2>&1
cp gpurun_out/variants/bench_mixed.txt $O/
MSPACK_HIP_NCHUNKS=1 MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace_mixed.so timeout 300 python tools/pipe_trace.py 4096 > $O/phases_mixed.txt 2>&1
MSPACK_HIP_SO=$R/build/variants/libmspack_hip_mixed.so timeout 900 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_lzx_frames.py tests/test_chm_extract.py > $O/parity.log 2>&1; echo "parity mixed rc=$?" >> $O/summary.txt; tail -n 2 $O/parity.log >> $O/summary.txt
cat $O/bench_mixed.txt; head -24 $O/phases_mixed.txt; cat $O/summary.txt
