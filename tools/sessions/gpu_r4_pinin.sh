#!/bin/bash
# round 4: the C drivers page-lock their input arenas (mspack_hip_pin): driver tests + through-API numbers
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pinin; mkdir -p $OUT; cd $R; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_api_bench.py tests/test_gpu_drivers.py tests/test_chm_extract.py tests/test_config2_cab.py tests/test_cabsets.py tests/test_chm_messages.py tests/test_gpu_messages.py tests/test_gpu_large_files.py tests/test_abi.py -q -x -m gpu 2>&1 | tail -3 ) | tee $OUT/tests.log
timeout 300 python tools/api_through.py 2 3 4 2>&1 | grep "^config" | tee $OUT/api.txt
