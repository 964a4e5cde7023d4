#!/bin/bash
# round 6, session Y (GPU box): the ticket order by the launch's shape (shipped default) -- launch-path parity with every order forced,
# the frame / CHM parity files, the launch shapes 512 .. 8192 against level order
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6y; mkdir -p $O
cd $R
timeout 1500 python -m pytest -m gpu -x -q -p no:cacheprovider tests/test_gpu_lzx_frames.py tests/test_chm_extract.py tests/test_gpu_lzx.py tests/test_gpu_fold.py tests/test_gpu_hostpath.py -k "not config5" > $O/parity.log 2>&1; echo "parity rc=$?" | tee -a $O/summary.txt; tail -n 2 $O/parity.log >> $O/summary.txt
B="python bench.py --exp --no-cpu --no-extras --steps 15 --warmup 4"
for rep in 1 2; do for u in 512 1024 4096 6144 8192; do for m in 0 3; do
  ( MSPACK_HIP_TICKET_ORDER=$m timeout 200 $B --units $u 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('order %d units %5d: ms_per_step %7.3f bit_exact %s' % ($m, $u, d['ms_per_step'], d['config']['bit_exact']))" ) >> $O/orders.txt 2>&1
done; done; done
timeout 600 python bench.py --host-path-worker --no-api 2> /dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('host path: to device %.3f ms %.1f MB/s   to host %.3f ms %.1f MB/s  bit_exact %s' % (d['ms'], d['MBps'], d['to_host_ms'], d['to_host_MBps'], d['bit_exact']))" >> $O/orders.txt
cat $O/summary.txt $O/orders.txt
