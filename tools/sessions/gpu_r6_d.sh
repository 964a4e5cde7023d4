#!/bin/bash
# round 6, session C (GPU box): the fold tasks' chain in two links (early / late gather) -- parity, the one-folder shapes, where a
# task's time goes (trace build), gathers in flight 32 / 48 / 60, non-temporal stores
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6d; mkdir -p $O
cd $R
T="tests/test_gpu_lzx_frames.py tests/test_gpu_mszip_blocks.py tests/test_gpu_runs.py tests/test_gpu_large_files.py"
MSPACK_HIP_FOLD=2 timeout 900 python -m pytest -m gpu -x -q -p no:cacheprovider $T -k "not launch_paths" > $O/parity_fold2.log 2>&1; echo "parity fold=2 rc=$?" | tee -a $O/summary.txt; tail -3 $O/parity_fold2.log >> $O/summary.txt
timeout 600 python tools/bench_folder_chain.py 4096 > $O/folder_chain.txt 2>&1; echo "folder chain rc=$?" >> $O/summary.txt
for v in ftrace w2 w8 w16; do
  echo "== variant $v" >> $O/variants.txt
  MSPACK_HIP_SO=$R/build/variants/libmspack_hip_$v.so timeout 300 python tools/fold_phases.py 512 >> $O/variants.txt 2>&1
  MSPACK_HIP_SO=$R/build/variants/libmspack_hip_$v.so timeout 300 python tools/fold_phases.py 512 32768 >> $O/variants.txt 2>&1
done
echo "== default build" >> $O/variants.txt
timeout 300 python tools/fold_phases.py 512 >> $O/variants.txt 2>&1
cat $O/summary.txt $O/folder_chain.txt $O/variants.txt
