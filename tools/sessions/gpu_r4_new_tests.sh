#!/bin/bash
# round 4: the new GPU tests (containers through the C harness, bench line incl. the strong-scaling branch, large-files prefix,
# config 5 at full size), then the through_api numbers
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4new; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_api_bench.py tests/test_gpu_bench_line.py tests/test_gpu_large_files.py "tests/test_gpu_hostpath.py::test_config5_shapes" -x -q -s -m gpu --durations=8 2>&1 | tail -40 ) > $OUT/pytest.log 2>&1
cat $OUT/pytest.log
