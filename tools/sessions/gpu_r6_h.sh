#!/bin/bash
# round 6, session H (GPU box): where the fold tasks pay -- n folders of f frames with the tasks off / by the rule / forced
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6h; mkdir -p $O
cd $R
for pol in 0 1 2; do MSPACK_HIP_FOLD=$pol timeout 900 python tools/fold_policy_sweep.py > $O/sweep_fold$pol.txt 2>&1; done
paste -d'\n' $O/sweep_fold0.txt $O/sweep_fold1.txt $O/sweep_fold2.txt | grep -v amdgpu.ids
