#!/bin/bash
# round 6, session AE (GPU box): the headline batch to the HOST (mspack_hip_decode_batch, 7.4 ms: H2D 1.7 + kernels 2.6 + D2H 4.8 side by
# side) -- the copy back is the long leg and cannot begin before the first chunk is through: chunk count x chunk shape x compute
# streams, with the chunks' hand-over times for the shipped setting and the best one
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6ae; mkdir -p $O
cd $R
for ncomp in 2 3 4; do
  for shape in 2 0 1 3; do
    echo "#### MSPACK_HIP_NCOMP_HOST=$ncomp MSPACK_HIP_CHUNK_SHAPE=$shape" >> $O/sweep.txt
    MSPACK_HIP_NCOMP_HOST=$ncomp MSPACK_HIP_CHUNK_SHAPE=$shape timeout 300 python tools/exp_hostpath.py 4096 5 4,6,8 2>&1 | grep -v to_device >> $O/sweep.txt
  done
done
echo "#### trace: shipped (4 chunks, shape 2, 2 streams)" >> $O/sweep.txt
MSPACK_HIP_TRACE=1 timeout 300 python tools/exp_hostpath.py 4096 2 4 2>&1 | grep "handed\|drain" | tail -n 5 >> $O/sweep.txt
echo "#### trace: 8 chunks, shape 3, 2 streams" >> $O/sweep.txt
MSPACK_HIP_TRACE=1 MSPACK_HIP_CHUNK_SHAPE=3 timeout 300 python tools/exp_hostpath.py 4096 2 8 2>&1 | grep "handed\|drain" | tail -n 9 >> $O/sweep.txt
cat $O/sweep.txt
