#!/bin/bash
# round 6, session N (GPU box): Quantum units with marks in a kernel of their own (mspack_decode_qtm_marks; the marks' state in LDS) --
# parity again, then config 4 without marks (must be 342 ms again) and with 64 marks per folder
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6n; mkdir -p $O
cd $R
timeout 1500 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_qtm.py tests/test_cab_sticky.py tests/test_gpu_hostpath.py tests/test_gpu_drivers.py -k "not config5" > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt; tail -5 $O/parity.log >> $O/summary.txt
timeout 300 python tools/bench_qtm_config4.py > $O/qtm.txt 2>&1
timeout 300 python tools/bench_qtm_config4.py 512 32 64 >> $O/qtm.txt 2>&1
timeout 300 python tools/bench_qtm_config4.py 128 32 >> $O/qtm.txt 2>&1
cat $O/summary.txt; grep kernel_ms $O/qtm.txt
