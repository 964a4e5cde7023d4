#!/bin/bash
# round 6, session S (GPU box): the compiler's scheduling strategies on the whole library (-mllvm -amdgpu-sched-strategy=max-ilp /
# max-memory-clause, -amdgpu-schedule-metric-bias=0) -- headline and 8192 units, two repetitions; Quantum config 4 with each
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6s; mkdir -p $O
cd $R
VARIANTS="schedilp schedmem bias0" REPS=1 TAG=sched bash tools/gpu_variants.sh > $O/variants.txt 2>&1
cp gpurun_out/variants/bench_sched.txt $O/
for v in schedilp schedmem bias0; do
  echo "$v: $(MSPACK_HIP_SO=$R/build/variants/libmspack_hip_$v.so timeout 300 python tools/bench_qtm_config4.py 2>&1 | grep kernel_ms)" >> $O/qtm.txt
done
cat $O/bench_sched.txt $O/qtm.txt
