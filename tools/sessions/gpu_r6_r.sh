#!/bin/bash
# round 6, session R (GPU box): randomized differential sweeps of the final build against the CPU oracle -- the frame-parallel LZX path and
# the block-parallel MSZIP path with the shipped launch rule, with the fold tasks forced on (MSPACK_HIP_FOLD=2) and off (=0), the streaming
# resolve off; the serial LZX kernel's sweep; every unit's error code, byte count, flags and bytes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6r; mkdir -p $O
cd $R
for fold in 1 2 0; do
  for seed in 61 62 63 64; do
    echo "MSPACK_HIP_FOLD=$fold tools/sweep_lzx_frames.py $seed 60: $(MSPACK_HIP_FOLD=$fold timeout 600 python tools/sweep_lzx_frames.py $seed 60 2>&1 | tail -n 1)" >> $O/sweeps.txt
    echo "MSPACK_HIP_FOLD=$fold tools/sweep_mszip_blocks.py $seed 40: $(MSPACK_HIP_FOLD=$fold timeout 600 python tools/sweep_mszip_blocks.py $seed 40 2>&1 | tail -n 1)" >> $O/sweeps.txt
  done
done
for seed in 65 66; do
  echo "MSPACK_HIP_STREAM_RESOLVE=0 tools/sweep_lzx_frames.py $seed 60: $(MSPACK_HIP_STREAM_RESOLVE=0 timeout 600 python tools/sweep_lzx_frames.py $seed 60 2>&1 | tail -n 1)" >> $O/sweeps.txt
  echo "tools/sweep_lzx.py $seed: $(timeout 600 python tools/sweep_lzx.py $seed 2>&1 | tail -n 1)" >> $O/sweeps.txt
  echo "tools/sweep_mszip.py $seed: $(timeout 600 python tools/sweep_mszip.py $seed 2>&1 | tail -n 1)" >> $O/sweeps.txt
done
cat $O/sweeps.txt
