#!/bin/bash
# round 4: the LDS-tile resolver with chain collapse as the resolve tasks' core: tile1 = everywhere, tile2 = many-frame units only
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4tile; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( MSPACK_HIP_SO=$R/build/variants/libmspack_hip_tile1.so timeout 600 python -m pytest tests/test_gpu_lzx_frames.py tests/test_gpu_kat.py tests/test_gpu_lzx.py -x -q -m gpu 2>&1 | tail -3 ) > $OUT/pytest_tile1.log 2>&1
VARIANTS="tile1" TAG=r4tile UNITS="4096 8192 1024" bash tools/gpu_variants.sh > /dev/null 2>&1
for v in base tile1; do
  so=$R/build/variants/libmspack_hip_$v.so; [ $v = base ] && so=$R/libmspack_amd/libmspack_hip.so
  ( echo "== $v"; MSPACK_HIP_SO=$so timeout 300 python tools/exp_bigfolder.py 512 child 2>&1 | tail -2 ) >> $OUT/bigfolder.txt 2>&1
done
cat $OUT/pytest_tile1.log gpurun_out/variants/bench_r4tile.txt $OUT/bigfolder.txt
