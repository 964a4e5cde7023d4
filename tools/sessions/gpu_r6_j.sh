#!/bin/bash
# round 6, session J (GPU box): streaming resolve decided from the launch's real ticket count -- parity of the frame tests, the launch
# shapes with the switch off and on
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6j; mkdir -p $O
cd $R
timeout 1500 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_lzx.py tests/test_gpu_lzx_frames.py tests/test_gpu_lzx_log.py tests/test_gpu_fuzz.py tests/test_chm_extract.py tests/test_chm_messages.py > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt; tail -3 $O/parity.log >> $O/summary.txt
for sw in 0 1 0 1; do
  MSPACK_HIP_STREAM_RESOLVE=$sw timeout 900 python bench.py --no-cpu --steps 20 --warmup 3 > $O/bench_stream$sw.json 2> $O/bench.err; echo "bench stream=$sw rc=$?" >> $O/summary.txt
  python - <<P
import json
d=json.loads(open("$O/bench_stream$sw.json").read().strip().splitlines()[-1])
print('stream', $sw, 'headline', d['ms_per_step'], d.get('value_host_inclusive'), d.get('value_host_to_host'))
for s in d.get('secondary', [])[:2] + d.get('secondary', [])[4:5]:
    print('   ', s['config'][:72], s.get('kernel_ms'), s.get('bit_exact'), (s.get('through_api') or {}).get('MBps'))
P
done
cat $O/summary.txt
