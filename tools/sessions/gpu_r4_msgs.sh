#!/bin/bash
# round 4: codec messages vs the reference's log, driver suites (the output arena is now shared by a batch's folders), the full bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4msgs; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_messages.py tests/test_gpu_drivers.py tests/test_cabsets.py tests/test_gpu_reference_suites.py tests/test_config2_cab.py tests/test_gpu_mszip.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -15 ) > $OUT/pytest.log 2>&1
( timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err )
cat $OUT/pytest.log; tail -c 6000 $OUT/bench.json; tail -3 $OUT/bench.err
