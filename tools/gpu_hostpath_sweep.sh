#!/bin/bash
# GPU box: host-buffer entry points, chunk count x compute-stream count
R=$GRAFT_REPO_ROOT
for ncomp in ${NCOMPUTE:-2 3 4}; do
  echo "#### MSPACK_HIP_NCOMPUTE=$ncomp"
  MSPACK_HIP_NCOMPUTE=$ncomp python $R/tools/exp_hostpath.py ${UNITS:-4096} 5 ${CHUNKS:-2,4,8}
done
