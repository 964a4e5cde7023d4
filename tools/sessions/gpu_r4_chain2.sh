#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4chain; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 300 python tools/exp_bigfolder.py 512 > $OUT/bigfolder2.txt 2>&1 )
( timeout 600 python -m pytest tests/test_gpu_lzx_frames.py tests/test_gpu_kat.py tests/test_gpu_drivers.py tests/test_gpu_large_files.py -x -q -s -m gpu 2>&1 | tail -8 ) > $OUT/pytest2.log 2>&1
tail -4 $OUT/bigfolder2.txt; cat $OUT/pytest2.log
