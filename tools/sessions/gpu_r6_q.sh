#!/bin/bash
# round 6, session Q (GPU box): write-through (sc1) stores for the parse waves' match records and / or literal rows -- an agent-scope
# release writes back every dirty line of the XCD's L2, so plain payload stores leave the L2 in pieces -- headline and 8192 units,
# parity of the frame tests, FETCH_SIZE / WRITE_SIZE of the headline launch per variant
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6q; mkdir -p $O
cd $R
VARIANTS="sc1r sc1w sc1a" REPS=1 TAG=sc1 bash tools/gpu_variants.sh > $O/variants.txt 2>&1
cp gpurun_out/variants/bench_sc1.txt $O/
for v in base sc1r sc1w sc1a; do
  so=$R/build/variants/libmspack_hip_$v.so; [ $v = base ] && so=$R/libmspack_amd/libmspack_hip.so
  rm -rf $R/gpurun_out/traffic
  MSPACK_HIP_SO=$so bash tools/gpu_traffic.sh > $O/traffic_$v.log 2>&1
  cp $R/gpurun_out/traffic/traffic.json $O/traffic_$v.json 2>/dev/null
  echo "== $v" >> $O/traffic.txt; grep "KiB per dispatch" $O/traffic_$v.log >> $O/traffic.txt
done
MSPACK_HIP_SO=$R/build/variants/libmspack_hip_sc1a.so timeout 900 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_lzx_frames.py tests/test_gpu_lzx.py tests/test_chm_extract.py -k "not launch_paths" > $O/parity_sc1a.log 2>&1; echo "parity sc1a rc=$?" >> $O/summary.txt; tail -n 2 $O/parity_sc1a.log >> $O/summary.txt
cat $O/bench_sc1.txt $O/traffic.txt $O/summary.txt
