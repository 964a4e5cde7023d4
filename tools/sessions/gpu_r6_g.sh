#!/bin/bash
# round 6, session G (GPU box): block headers read ahead of the header chain (lzx_pipe_spec_header), the parse task's tail and the
# speculation as calls of the ticket loop -- parity, the one-folder shapes, the headline with and without
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6g; mkdir -p $O
cd $R
timeout 1500 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_kat.py tests/test_gpu_lzx.py tests/test_gpu_lzx_frames.py tests/test_gpu_lzx_log.py tests/test_gpu_fold.py tests/test_gpu_fuzz.py tests/test_gpu_large_files.py tests/test_chm_extract.py tests/test_gpu_drivers.py > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt; tail -3 $O/parity.log >> $O/summary.txt
timeout 600 python tools/bench_folder_chain.py 4096 > $O/folder_chain.txt 2>&1; echo "folder chain rc=$?" >> $O/summary.txt
MSPACK_HIP_SO=$R/build/variants/libmspack_hip_nospec.so timeout 600 python tools/bench_folder_chain.py 4096 > $O/folder_chain_nospec.txt 2>&1
for i in 1 2; do
  timeout 600 python bench.py --no-cpu --no-extras --steps 20 --warmup 3 > $O/bench_spec_$i.json 2> $O/bench.err
  MSPACK_HIP_SO=$R/build/variants/libmspack_hip_nospec.so timeout 600 python bench.py --no-cpu --no-extras --steps 20 --warmup 3 > $O/bench_nospec_$i.json 2>> $O/bench.err
done
cat $O/summary.txt $O/folder_chain.txt; echo "--- without the speculation:"; cat $O/folder_chain_nospec.txt
python - <<P
import json
for n in ("spec_1","nospec_1","spec_2","nospec_2"):
    d=json.loads(open("$O/bench_%s.json" % n).read().strip().splitlines()[-1]); print(n, d['ms_per_step'], d['value'])
P
