#!/bin/bash
# round 6, session B (GPU box): the fold path (mspack_lzx_fold / mspack_mszip_fold) on the hardware -- parity with the default
# rule and with the path forced, the one-folder shapes with it off and on, the kernels' split, the headline beside it
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6b; mkdir -p $O
cd $R
T="tests/test_gpu_lzx_frames.py tests/test_gpu_mszip_blocks.py tests/test_gpu_runs.py tests/test_gpu_large_files.py"
timeout 900 python -m pytest -m gpu -x -q -p no:cacheprovider $T tests/test_gpu_lzx.py tests/test_gpu_mszip.py > $O/parity_default.log 2>&1; echo "parity default rc=$?" | tee -a $O/summary.txt; tail -3 $O/parity_default.log >> $O/summary.txt
MSPACK_HIP_FOLD=2 timeout 900 python -m pytest -m gpu -x -q -p no:cacheprovider $T -k "not launch_paths" > $O/parity_fold2.log 2>&1; echo "parity fold=2 rc=$?" | tee -a $O/summary.txt; tail -3 $O/parity_fold2.log >> $O/summary.txt
for pol in 0 1; do
  MSPACK_HIP_FOLD=$pol timeout 600 python tools/bench_folder_chain.py 4096 > $O/folder_chain_fold$pol.txt 2>&1; echo "folder chain fold=$pol rc=$?" >> $O/summary.txt
done
cd /tmp; export TMPDIR=/tmp
MSPACK_HIP_FOLD=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fold -o fc -- python $R/tools/bench_folder_chain.py 4096 > $O/stats_fold.log 2>&1
for f in $(find $O/stats_fold -name "*kernel_stats.csv"); do cp $f $O/folder_chain_kernel_stats.csv; done
rm -rf $O/stats_fold
cd $R
timeout 600 python bench.py --no-cpu --no-extras --steps 20 --warmup 3 > $O/bench_headline.json 2> $O/bench_headline.err; echo "bench rc=$?" >> $O/summary.txt
cat $O/summary.txt; cat $O/folder_chain_fold0.txt $O/folder_chain_fold1.txt; head -12 $O/folder_chain_kernel_stats.csv | cut -c1-150
python -c "
import json; d=json.loads(open('$O/bench_headline.json').read().strip().splitlines()[-1]); print('headline', d['ms_per_step'], d['value'])"
