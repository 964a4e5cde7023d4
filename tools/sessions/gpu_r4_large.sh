#!/bin/bash
# round 4: the reference's large-files.test at full size (three folders of 65 535 blocks = 2 GiB each), timings + device memory
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/large; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( while true; do rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used" | head -1; sleep 5; done ) > $OUT/vram.log 2>&1 &
MON=$!
( MSPACK_TEST_LARGE=1 timeout 420 python tests/test_gpu_large_files.py 2>&1 | grep -v amdgpu.ids ) > $OUT/large.txt 2>&1
kill $MON
cat $OUT/large.txt; sort -t: -k3 -n $OUT/vram.log | tail -1
