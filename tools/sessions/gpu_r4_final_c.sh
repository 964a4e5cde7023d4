#!/bin/bash
# round 4, final validation C: per-phase sums and per-ticket timeline of ONE 4096-unit launch (trace build, one chunk)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4f; mkdir -p $OUT; cd $R
( MSPACK_HIP_NCHUNKS=1 MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace.so timeout 200 python tools/pipe_trace.py 4096 > $OUT/phases4096.txt 2>&1 )
( MSPACK_HIP_NCHUNKS=1 MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace.so timeout 200 python tools/pipe_trace.py 8192 > $OUT/phases8192.txt 2>&1 )
tail -22 $OUT/phases4096.txt; tail -9 $OUT/phases8192.txt
