#!/bin/bash
# round 4: what a single long folder costs -- one 512-frame LZX folder through cabd->extract(), MSZIP folders (2 x 2000 blocks, 512 x 8)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4chain; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 300 python tools/exp_bigfolder.py 512 > $OUT/bigfolder.txt 2>&1 )
( timeout 300 python tools/bench_mszip_folder.py 2 2000 > $OUT/mszip_2x2000.txt 2>&1 )
( timeout 300 python tools/bench_mszip_folder.py 512 8 > $OUT/mszip_512x8.txt 2>&1 )
cat $OUT/bigfolder.txt; tail -4 $OUT/mszip_2x2000.txt; tail -4 $OUT/mszip_512x8.txt
