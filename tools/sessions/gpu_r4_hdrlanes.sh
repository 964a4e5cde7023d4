#!/bin/bash
# round 4: block headers through the lane parser (lzx_read_lens_lanes) -- A/B: base (a real call), hdrinl (inlined), hdrold (-DLZX_NO_HDR_LANES)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/hdrlanes; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
rm -f gpurun_out/variants/bench_hdr.txt
TAG=hdr VARIANTS="hdrinl hdrold" REPS=2 UNITS="4096 8192 1024" bash tools/gpu_variants.sh > /dev/null 2>&1
cat gpurun_out/variants/bench_hdr.txt
( MSPACK_HIP_NCHUNKS=1 MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace.so timeout 200 python tools/pipe_trace.py 4096 2>&1 | head -14 )
