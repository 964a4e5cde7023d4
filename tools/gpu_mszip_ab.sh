#!/bin/bash
# GPU box: MSZIP launch shapes, the shipped library against analysis builds (VARIANTS: names under build/variants/)
R=$GRAFT_REPO_ROOT
for v in base ${VARIANTS:-chain}; do
  so=$R/libmspack_amd/libmspack_hip.so; [ "$v" != base ] && so=$R/build/variants/libmspack_hip_$v.so
  for shape in "4096 1" "512 8" "2 2000"; do
    echo "== $v: $shape"
    MSPACK_HIP_SO=$so python $R/tools/bench_mszip_folder.py $shape 2>&1 | grep -v Warning
  done
done
