#!/bin/bash
# round 6, session V (GPU box): config 4 (and 2, 3) through the object API once the input arena has room for the marks' tables
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6v; mkdir -p $O
cd $R
timeout 600 python tools/api_through.py 4 > $O/api.txt 2>&1
timeout 600 python tools/api_through.py 2 3 >> $O/api.txt 2>&1
timeout 600 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_cab_sticky.py tests/test_api_bench.py tests/test_config2_cab.py >> $O/parity.log 2>&1; echo "parity rc=$?" >> $O/api.txt
cat $O/api.txt
