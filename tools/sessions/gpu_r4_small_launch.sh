#!/bin/bash
# round 4: uniform launches with no more frame slots than waves take the merged (one ticket per frame) order again
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; export TMPDIR=/tmp
( timeout 200 python -m pytest tests/test_gpu_lzx_frames.py tests/test_chm_extract.py tests/test_api_bench.py tests/test_gpu_lzx_log.py "tests/test_gpu_hostpath.py::test_headline_batch_host_entry_points" -q -x -m gpu -k "not launch_paths" 2>&1 | tail -2 )
for u in 512 1024 1365 4096; do timeout 100 python bench.py --exp --no-cpu --no-extras --steps 15 --warmup 4 --units $u 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('units %5d: ms_per_step %7.3f bit_exact %s adopted %s' % ($u, d['ms_per_step'], d['config']['bit_exact'], d['config']['units_on_frame_parallel_path']))"; done
timeout 100 python bench.py --host-path-worker --units 4096 --unit-kib 64 --text 0 --no-api 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('host_inclusive', d.get('MBps'), d.get('ms'), d.get('to_host_MBps'), d.get('to_host_ms'))"
