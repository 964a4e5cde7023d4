#!/bin/bash
# round 6, session AD (GPU box): config 2 through the object API with the cabinet driver's jobs (the parts' index in), jobs off / on;
# the cabinet parity files; the default bench line; which tests of the suite take the time
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6ad; mkdir -p $O
cd $R
for j in 0 1 0 1; do
  echo "== MSPACK_HIP_JOBS=$j" >> $O/api.txt
  MSPACK_HIP_JOBS=$j timeout 600 python tools/api_through.py 2 >> $O/api.txt 2>&1
done
echo "== trace, config 2" >> $O/api.txt
MSPACK_HIP_TRACE=1 timeout 600 python tools/api_through.py 2 2>&1 | grep -v "^config" | tail -n 6 >> $O/api.txt
cut -c1-420 $O/api.txt
timeout 1500 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_drivers.py tests/test_cab_sticky.py tests/test_cabsets.py tests/test_config2_cab.py tests/test_api_bench.py tests/test_gpu_messages.py tests/test_gpu_reference_suites.py > $O/parity.log 2>&1; echo "cabinet parity rc=$?" | tee -a $O/summary.txt; tail -n 2 $O/parity.log >> $O/summary.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python - <<P
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','value_host_inclusive','value_host_to_host')})
for s in d['secondary']:
    print(s['config'][:60], '|', s.get('kernel_ms'), s.get('value'), s.get('bit_exact'), (s.get('through_api') or {}).get('MBps'))
P
timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 -k "hostpath or config5 or large or reference_suites or fuzz or qtm or fold" > $O/durations.txt 2>&1; tail -n 32 $O/durations.txt
cat $O/summary.txt
