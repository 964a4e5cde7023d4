#!/bin/bash
# round 6, session P (GPU box): the host path's chunk launches with the streaming resolve too (MSPACK_HIP_STREAM_RESOLVE=2) against the
# shipped rule (only launches that run alone) -- bench.py's clean-process worker, headline batch, two repetitions
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6p; mkdir -p $O
cd $R
for rep in 1 2; do for sw in 1 2; do
  MSPACK_HIP_STREAM_RESOLVE=$sw timeout 600 python bench.py --host-path-worker --no-api 2> $O/err_$sw.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('stream=$sw: to device %.3f ms %.1f MB/s   to host %.3f ms %.1f MB/s  bit_exact %s' % (d['ms'], d['MBps'], d['to_host_ms'], d['to_host_MBps'], d['bit_exact']))" >> $O/hostpath.txt 2>&1
done; done
cat $O/hostpath.txt
