#!/bin/bash
# round 6, session O (GPU box): how much of the lane parser's count walks are the walks behind the second (trace build's counters)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6o; mkdir -p $O
cd $R
MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace.so timeout 300 python tools/pipe_trace.py 4096 > $O/trace.txt 2>&1
head -16 $O/trace.txt
