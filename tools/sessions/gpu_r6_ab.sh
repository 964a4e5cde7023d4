#!/bin/bash
# round 6, session AB (GPU box): config 3 through the object API on jobs -- how many compute streams the chunks' launches of a batch
# that goes back to the host run on (MSPACK_HIP_NCOMP_HOST: 2 was measured on the headline batch, whose chunks each fill the chip),
# with the chunks' hand-over times; the headline's host path with the same knob
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6ab; mkdir -p $O
cd $R
for nc in 2 4; do
  for sh in 2 0; do
    echo "== MSPACK_HIP_NCOMP_HOST=$nc MSPACK_HIP_CHUNK_SHAPE=$sh" >> $O/api.txt
    MSPACK_HIP_NCOMP_HOST=$nc MSPACK_HIP_CHUNK_SHAPE=$sh timeout 600 python tools/api_through.py 3 >> $O/api.txt 2>&1
    MSPACK_HIP_NCOMP_HOST=$nc MSPACK_HIP_CHUNK_SHAPE=$sh MSPACK_HIP_TRACE=1 timeout 600 python tools/api_through.py 3 2>&1 | grep -v "^config" | tail -n 5 >> $O/api.txt
  done
done
for nc in 2 4; do
  echo "== headline, host path, MSPACK_HIP_NCOMP_HOST=$nc" >> $O/api.txt
  MSPACK_HIP_NCOMP_HOST=$nc timeout 600 python tools/bench_hostpath.py 2>&1 | tail -n 6 >> $O/api.txt
done
cat $O/api.txt
