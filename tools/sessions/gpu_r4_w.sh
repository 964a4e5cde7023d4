#!/bin/bash
# round 4: persistent waves per CU of mspack_lzx_pipe (16 = what the LDS allows) -- how much do fewer cost?
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4w; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
for w in 16 12 8; do for n in 4096 8192; do
  ( MSPACK_HIP_PIPE_WAVES_PER_CU=$w timeout 300 python bench.py --no-cpu --no-extras --units $n --steps 10 --warmup 3 --exp > $OUT/b_${w}_$n.json 2>> $OUT/bench.err )
  python - <<PY
import json
try:
    j = json.loads(open("$OUT/b_${w}_$n.json").read().strip().splitlines()[-1]); print("waves/CU $w units $n ms", j["ms_per_step"])
except Exception as e: print("ERR", e)
PY
done; done
