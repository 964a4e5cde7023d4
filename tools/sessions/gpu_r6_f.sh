#!/bin/bash
# round 6, session F (GPU box): the new fold tests, the whole suite minus the five-minute shapes (SIGABRT watch: tests/conftest.py keeps
# the native backtrace), Quantum with and without its output side (the floor of a producer / consumer split)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6f; mkdir -p $O
cd $R
timeout 900 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_fold.py > $O/fold_tests.log 2>&1; echo "fold tests rc=$?" | tee -a $O/summary.txt; tail -3 $O/fold_tests.log >> $O/summary.txt
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not config5 and not large_files and not launch_paths_same_bytes and not test_gpu_fold" > $O/suite.log 2>&1; echo "suite rc=$?" | tee -a $O/summary.txt; tail -3 $O/suite.log >> $O/summary.txt
cp tests/_build/abort_bt.log $O/ 2>/dev/null
timeout 300 python tools/bench_qtm_config4.py > $O/qtm_shipped.txt 2>&1
MSPACK_HIP_SO=$R/build/variants/libmspack_hip_qtm_noout.so timeout 300 python tools/bench_qtm_config4.py > $O/qtm_noout.txt 2>&1
MSPACK_HIP_SO=$R/build/variants/libmspack_hip_qtm_noout.so timeout 300 python tools/bench_qtm_config4.py 128 32 >> $O/qtm_noout.txt 2>&1
timeout 300 python tools/bench_qtm_config4.py 128 32 >> $O/qtm_shipped.txt 2>&1
cat $O/summary.txt; echo "--- shipped"; cat $O/qtm_shipped.txt; echo "--- no output"; cat $O/qtm_noout.txt
