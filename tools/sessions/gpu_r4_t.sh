#!/bin/bash
# round 4: per-ticket timeline of ONE 4096-unit launch (host path forced to one chunk), trace build
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4t; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( MSPACK_HIP_NCHUNKS=1 MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace.so timeout 200 python tools/pipe_trace.py 4096 $OUT/trace4096.npy > $OUT/phases4096.txt 2>&1 )
( MSPACK_HIP_NCHUNKS=1 MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace.so timeout 200 python tools/pipe_trace.py 8192 > $OUT/phases8192.txt 2>&1 )
tail -12 $OUT/phases4096.txt; tail -9 $OUT/phases8192.txt
