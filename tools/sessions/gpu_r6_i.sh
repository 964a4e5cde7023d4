#!/bin/bash
# round 6, session I (GPU box): resolve tasks that take their frames up while they are parsed (launches with a wave for every ticket):
# parity, then the launch shapes with the switch off and on
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6i; mkdir -p $O
cd $R
timeout 1500 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_kat.py tests/test_gpu_lzx.py tests/test_gpu_lzx_frames.py tests/test_gpu_lzx_log.py tests/test_gpu_fold.py tests/test_gpu_fuzz.py tests/test_chm_extract.py tests/test_gpu_hostpath.py -k "not config5" > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt; tail -3 $O/parity.log >> $O/summary.txt
for sw in 0 1; do
  MSPACK_HIP_STREAM_RESOLVE=$sw timeout 900 python bench.py --no-cpu --steps 20 --warmup 3 > $O/bench_stream$sw.json 2> $O/bench.err; echo "bench stream=$sw rc=$?" >> $O/summary.txt
done
cat $O/summary.txt
python - <<P
import json
for sw in (0, 1):
    d=json.loads(open("$O/bench_stream%d.json" % sw).read().strip().splitlines()[-1])
    print('stream', sw, 'headline', d['ms_per_step'], d.get('value_host_inclusive'), d.get('value_host_to_host'))
    for s in d.get('secondary', []):
        print('   ', s['config'][:72], s.get('kernel_ms'), s.get('bit_exact'), (s.get('through_api') or {}).get('MBps'))
P
