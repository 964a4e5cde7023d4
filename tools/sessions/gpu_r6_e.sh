#!/bin/bash
# round 6, session E (GPU box): fold tasks with eight waves and the chain's gathers in three steps; units of long runs keep the
# pipe's resolve tasks; MSZIP's in_next; the new bench secondaries
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6e; mkdir -p $O
cd $R
T="tests/test_gpu_lzx_frames.py tests/test_gpu_mszip_blocks.py tests/test_gpu_runs.py tests/test_gpu_large_files.py"
MSPACK_HIP_FOLD=2 timeout 900 python -m pytest -m gpu -x -q -p no:cacheprovider $T -k "not launch_paths" > $O/parity_fold2.log 2>&1; echo "parity fold=2 rc=$?" | tee -a $O/summary.txt; tail -3 $O/parity_fold2.log >> $O/summary.txt
timeout 900 python -m pytest -m gpu -x -q -p no:cacheprovider $T tests/test_gpu_mszip.py tests/test_cab_sticky.py tests/test_gpu_drivers.py tests/test_gpu_hostpath.py -k "not config5" > $O/parity_default.log 2>&1; echo "parity default rc=$?" | tee -a $O/summary.txt; tail -3 $O/parity_default.log >> $O/summary.txt
timeout 600 python tools/bench_folder_chain.py 4096 > $O/folder_chain.txt 2>&1; echo "folder chain rc=$?" >> $O/summary.txt
MSPACK_HIP_FOLD=2 timeout 600 python tools/bench_folder_chain.py 4096 > $O/folder_chain_fold2.txt 2>&1
for v in ftrace w4 w16; do
  echo "== variant $v" >> $O/variants.txt
  MSPACK_HIP_SO=$R/build/variants/libmspack_hip_$v.so timeout 300 python tools/fold_phases.py 512 >> $O/variants.txt 2>&1
done
echo "== default build" >> $O/variants.txt
timeout 300 python tools/fold_phases.py 512 >> $O/variants.txt 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
cat $O/summary.txt $O/folder_chain.txt; echo "--- fold forced:"; cat $O/folder_chain_fold2.txt; cat $O/variants.txt
python - <<P
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print('headline', d['ms_per_step'], d['value'], d.get('value_host_inclusive'), d.get('value_host_to_host'))
for s in d.get('secondary', []):
    print(s['config'][:70], s.get('kernel_ms'), s.get('value'), s.get('bit_exact'), s.get('units_on_frame_parallel_path'), (s.get('cpu_baseline') or {}).get('value'), (s.get('cpu_baseline') or {}).get('cores'))
P
