#!/bin/bash
# round 4: Quantum -- parity tests, then config 4 with the round-3 kernel (variant qtm_r3) and the new one on the same box
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4qtm; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_qtm.py tests/test_gpu_fuzz.py tests/test_gpu_messages.py -x -q -m gpu 2>&1 | tail -8 ) > $OUT/pytest.log 2>&1
( MSPACK_HIP_SO=$R/build/variants/libmspack_hip_qtm_r3.so timeout 300 python tools/bench_qtm_config4.py 2>&1 | tail -1 ) > $OUT/bench.txt 2>&1
( timeout 300 python tools/bench_qtm_config4.py 2>&1 | tail -1 ) >> $OUT/bench.txt 2>&1
( MSPACK_HIP_SO=$R/build/variants/libmspack_hip_qtm_r3.so timeout 300 python tools/bench_qtm_config4.py 512 4 2>&1 | tail -1 ) >> $OUT/bench.txt 2>&1
( timeout 300 python tools/bench_qtm_config4.py 512 4 2>&1 | tail -1 ) >> $OUT/bench.txt 2>&1
cat $OUT/pytest.log $OUT/bench.txt
