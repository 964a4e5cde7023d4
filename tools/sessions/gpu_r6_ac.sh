#!/bin/bash
# round 6, session AC (GPU box): the cabinet driver on jobs too -- the drivers' parity files, configs 2 / 3 / 4 through the object
# API with jobs off and on, then the whole -m gpu suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6ac; mkdir -p $O
cd $R
timeout 1500 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_hostpath.py tests/test_chm_extract.py tests/test_gpu_drivers.py tests/test_cab_sticky.py tests/test_cabsets.py tests/test_config2_cab.py tests/test_api_bench.py tests/test_gpu_messages.py tests/test_gpu_reference_suites.py -k "not config5" > $O/parity.log 2>&1; echo "drivers' parity rc=$?" | tee -a $O/summary.txt; tail -n 3 $O/parity.log >> $O/summary.txt
for j in 0 1; do
  echo "== MSPACK_HIP_JOBS=$j" >> $O/api.txt
  MSPACK_HIP_JOBS=$j timeout 600 python tools/api_through.py 2 3 >> $O/api.txt 2>&1
  MSPACK_HIP_JOBS=$j timeout 600 python tools/api_through.py 2 3 >> $O/api.txt 2>&1
  MSPACK_HIP_JOBS=$j timeout 600 python tools/api_through.py 4 >> $O/api.txt 2>&1
done
echo "== trace, config 2 and 3" >> $O/api.txt
MSPACK_HIP_TRACE=1 timeout 600 python tools/api_through.py 2 3 2>&1 | grep -v "^config" | tail -n 12 >> $O/api.txt
cut -c1-420 $O/api.txt
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; echo "whole suite rc=$?" | tee -a $O/summary.txt; tail -n 3 $O/pytest_gpu.txt >> $O/summary.txt
cat $O/summary.txt
