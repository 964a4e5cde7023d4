#!/bin/bash
# round 6, session L (GPU box): where a parse task's time goes with the first walk's tail at 384 and at 192 bits -- steps of the count
# walks, rounds and steps of the last walk per pass (trace build's counters), the phases' times
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6l; mkdir -p $O
cd $R
for v in trace trace_tail192; do
  echo "== $v" >> $O/trace.txt
  MSPACK_HIP_SO=$R/build/variants/libmspack_hip_$v.so timeout 300 python tools/pipe_trace.py 4096 >> $O/trace.txt 2>&1
done
cat $O/trace.txt
