#!/bin/bash
# round 6, session T (GPU box): config 3's launch shape on the final build, streaming resolve off / on, with and without the CPU legs
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6t; mkdir -p $O
cd $R
for sw in 0 1 0 1; do
  MSPACK_HIP_STREAM_RESOLVE=$sw timeout 900 python bench.py --no-cpu --steps 10 --warmup 3 > $O/b.json 2> $O/b.err
  python - <<P >> $O/ab.txt
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print('stream', $sw, 'no-cpu: headline', d['ms_per_step'], [(s['config'][:28], s.get('kernel_ms')) for s in d['secondary'][:2] + d['secondary'][4:5]])
P
done
timeout 900 python bench.py --steps 10 --warmup 3 > $O/b.json 2> $O/b.err
python - <<P >> $O/ab.txt
import json
d=json.loads(open("$O/b.json").read().strip().splitlines()[-1])
print('default, with cpu: headline', d['ms_per_step'], [(s['config'][:28], s.get('kernel_ms')) for s in d['secondary'][:2] + d['secondary'][4:5]])
P
cat $O/ab.txt
