#!/bin/bash
# round 6, session X (GPU box): the ticket order of a uniform launch -- level order / mixed sections / unit-major (MSPACK_HIP_TICKET_ORDER
# 0 / 1 / 2) at 512 .. 8192 intervals, two repetitions
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6x; mkdir -p $O
cd $R
B="python bench.py --exp --no-cpu --no-extras --steps 15 --warmup 4"
for rep in 1 2; do for u in 512 1024 2048 3072 4096 6144 8192; do for m in 0 1 2; do
  ( MSPACK_HIP_TICKET_ORDER=$m timeout 200 $B --units $u 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('order %d units %5d: ms_per_step %7.3f bit_exact %s adopted %s' % ($m, $u, d['ms_per_step'], d['config']['bit_exact'], d['config']['units_on_frame_parallel_path']))
except Exception as e: print('order $m units $u: FAILED', e)
" ) >> $O/orders.txt 2>&1
done; done; done
cat $O/orders.txt
