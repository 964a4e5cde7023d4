#!/bin/bash
# round 6, session U (GPU box): config 4 through the object API after the marks' file walk became one pass; the compiler's scheduling
# strategies (session S's script); the randomized sweeps of the final build (session R's script)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6u; mkdir -p $O
cd $R
timeout 600 python tools/api_through.py 4 3 2 > $O/api.txt 2>&1; cat $O/api.txt
timeout 900 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_cab_sticky.py tests/test_gpu_drivers.py tests/test_api_bench.py > $O/parity.log 2>&1; echo "parity rc=$?"; tail -n 2 $O/parity.log
bash tools/sessions/gpu_r6_s.sh > $O/s.log 2>&1; tail -n 22 $O/s.log
bash tools/sessions/gpu_r6_r.sh > $O/r.log 2>&1; tail -n 36 $O/r.log
