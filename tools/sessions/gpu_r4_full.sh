#!/bin/bash
# round 4: smoke, the whole -m gpu suite, HBM-side traffic of the headline launch (separate --pmc passes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4full; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -3 ) > $OUT/smoke.log 2>&1
( timeout 1800 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -16 ) > $OUT/pytest.log 2>&1
( timeout 400 bash tools/gpu_traffic.sh > $OUT/traffic_lzx.txt 2>&1; cp gpurun_out/traffic/traffic.json $OUT/traffic_lzx.json )
cat $OUT/smoke.log $OUT/pytest.log; cat $OUT/traffic_lzx.json
