#!/bin/bash
# round 3, final validation A: smoke(), the whole -m gpu suite, the default bench line (what the driver runs)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r3f; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -3 ) > $OUT/smoke.log 2>&1
( timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 2>&1 | tail -16 ) > $OUT/pytest.log 2>&1
( timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err )
cat $OUT/smoke.log $OUT/pytest.log; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err
