#!/bin/bash
# round 6, session AA (GPU box): jobs (mspack_hip_decode_batch_begin) -- parity of the new entry points and of the CHM driver on
# them, then config 3 (and 2) through the object API with jobs off / on and with the chunk count varied
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6aa; mkdir -p $O
cd $R
timeout 1200 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_hostpath.py tests/test_chm_extract.py tests/test_chmdir.py tests/test_api_bench.py tests/test_gpu_reference_suites.py -k "not config5" > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt; tail -n 3 $O/parity.log >> $O/summary.txt
for j in 0 1; do
  echo "== MSPACK_HIP_JOBS=$j" >> $O/api.txt
  MSPACK_HIP_JOBS=$j timeout 600 python tools/api_through.py 3 >> $O/api.txt 2>&1
  MSPACK_HIP_JOBS=$j timeout 600 python tools/api_through.py 3 >> $O/api.txt 2>&1
done
for nc in 2 6 8; do
  echo "== jobs on, MSPACK_HIP_NCHUNKS=$nc" >> $O/api.txt
  MSPACK_HIP_NCHUNKS=$nc MSPACK_HIP_CHUNK_UNITS=64 timeout 600 python tools/api_through.py 3 >> $O/api.txt 2>&1
done
echo "== jobs on, chunk shape 0 / 1 (4 chunks)" >> $O/api.txt
MSPACK_HIP_CHUNK_SHAPE=0 timeout 600 python tools/api_through.py 3 >> $O/api.txt 2>&1
MSPACK_HIP_CHUNK_SHAPE=1 timeout 600 python tools/api_through.py 3 >> $O/api.txt 2>&1
echo "== trace" >> $O/api.txt
MSPACK_HIP_TRACE=1 timeout 600 python tools/api_through.py 3 2>&1 | tail -n 8 >> $O/api.txt
echo "== config 2 (the cabinet driver: no jobs yet)" >> $O/api.txt
timeout 600 python tools/api_through.py 2 >> $O/api.txt 2>&1
cat $O/api.txt $O/summary.txt
