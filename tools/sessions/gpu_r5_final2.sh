#!/bin/bash
# round 5, the last GPU action: smoke + the whole -m gpu suite as the driver runs it, at the final commit
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5final2; mkdir -p $OUT; cd $R
python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 740 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=6 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.log
