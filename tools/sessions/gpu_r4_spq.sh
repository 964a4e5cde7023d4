#!/bin/bash
# round 4: resolve span / queue capacity variants (SPQ_CAP, SPQ_SPAN) against the shipped 160 / 512
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
rm -f gpurun_out/variants/bench_spq.txt
TAG=spq VARIANTS="s768 q224 q288" REPS=2 UNITS="4096 8192" bash tools/gpu_variants.sh > /dev/null 2>&1
cat gpurun_out/variants/bench_spq.txt
for v in base q224; do so=$R/build/variants/libmspack_hip_$v.so; [ $v = base ] && so=$R/libmspack_amd/libmspack_hip.so
  echo "== mszip $v"; MSPACK_HIP_SO=$so python tools/bench_mszip_folder.py 512 8 2>&1 | grep block_parse.*True | cut -c1-200; done
