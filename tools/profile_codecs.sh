#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 --kernel-trace --stats for the MSZIP and Quantum configurations
# (BASELINE.json configs 2 and 4, tools/bench_codecs.py); outputs under gpurun_out/profile_codecs/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile_codecs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in mszip qtm; do
  PYTHONPATH=$R:$R/tests rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$c -o trace -- python $R/tools/bench_codecs.py $c > $OUT/$c.log 2>&1
  echo "== $c"; tail -1 $OUT/$c.log
  for f in $(find $OUT/$c -name '*kernel_stats.csv'); do head -4 $f; done
done
