#!/bin/bash
# round 4, session 6: the new message tests, the drivers' gather change, huge-page advice on/off
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/s6
timeout 900 python -m pytest tests/test_chm_messages.py tests/test_gpu_lzx_log.py tests/test_gpu_messages.py tests/test_api_bench.py tests/test_config2_cab.py tests/test_cabsets.py tests/test_chm_extract.py -m gpu -x -q > gpurun_out/s6/tests.log 2>&1
tail -3 gpurun_out/s6/tests.log
cat /sys/kernel/mm/transparent_hugepage/enabled
MSPACK_ARENA_HUGEPAGES=0 timeout 600 python tools/api_through.py 2 3 2>&1 | tail -2 | tee gpurun_out/s6/api_off.log
timeout 600 python tools/api_through.py 2 3 2>&1 | tail -2 | tee gpurun_out/s6/api_on.log
