#!/bin/bash
# round 4, last session: the final build (MSZIP literal ring in) -- smoke, the -m gpu suite without the 4-minute full config-5 case
# (it ran in gpu_r4_final_a.sh on a build whose LZX kernels are the same), the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4g; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -3 ) > $OUT/smoke.log 2>&1
( timeout 1500 python -m pytest tests -q -m gpu -k "not full65536" --durations=5 2>&1 | tail -12 ) > $OUT/pytest.log 2>&1
( timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err )
cat $OUT/smoke.log $OUT/pytest.log; tail -c 700 $OUT/bench.json; tail -3 $OUT/bench.err
