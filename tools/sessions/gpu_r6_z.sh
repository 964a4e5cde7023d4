#!/bin/bash
# round 6, session Z (GPU box): idle page-locked blocks trimmed oldest first -- the staging-pool and lifetime tests, then the default bench
# line (config 4 through the API behind configs 2 and 3 in one process, as bench.py runs them)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6z; mkdir -p $O
cd $R
timeout 900 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_hostpath.py tests/test_api_bench.py tests/test_config2_cab.py tests/test_chm_extract.py -k "not config5" > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt; tail -n 2 $O/parity.log >> $O/summary.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<P
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_host_inclusive','value_host_to_host')})
for s in d['secondary']:
    print(s['config'][:60], '|', s.get('kernel_ms'), s.get('value'), s.get('bit_exact'), (s.get('through_api') or {}).get('MBps'))
P
cat $O/summary.txt
