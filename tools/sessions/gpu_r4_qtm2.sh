#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4qtm; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_qtm.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3 ) > $OUT/pytest2.log 2>&1
( MSPACK_HIP_SO=$R/build/variants/libmspack_hip_qtm_r3.so timeout 300 python tools/bench_qtm_config4.py 2>&1 | tail -1 ) > $OUT/bench2.txt 2>&1
( timeout 300 python tools/bench_qtm_config4.py 2>&1 | tail -1 ) >> $OUT/bench2.txt 2>&1
( MSPACK_HIP_SO=$R/build/variants/libmspack_hip_qtm_tm.so timeout 300 python tools/bench_qtm_config4.py 512 4 2>&1 | tail -3 ) >> $OUT/bench2.txt 2>&1
cat $OUT/pytest2.log $OUT/bench2.txt
