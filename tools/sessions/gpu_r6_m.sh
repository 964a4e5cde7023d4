#!/bin/bash
# round 6, session M (GPU box): MSPACK_HIP_UF_QTM_MARKS on the hardware -- the marks test, the cabinet goldens of requests that hold
# bytes back (real cabd's answers), the host path's mixed batches with marks, the rest of the Quantum / driver tests; Quantum's
# config 4 timing with the compare per token in the loop
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6m; mkdir -p $O
cd $R
timeout 1500 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_qtm.py tests/test_cab_sticky.py tests/test_gpu_hostpath.py tests/test_gpu_drivers.py tests/test_gpu_reference_suites.py -k "not config5" > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt; tail -5 $O/parity.log >> $O/summary.txt
timeout 300 python tools/bench_qtm_config4.py > $O/qtm.txt 2>&1
timeout 300 python tools/bench_qtm_config4.py 128 32 >> $O/qtm.txt 2>&1
cat $O/summary.txt; cat $O/qtm.txt
