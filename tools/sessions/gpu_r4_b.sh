#!/bin/bash
# round 4, quick perf loop: headline / 8192 / 1024 units (no parity gate beyond bit_exact of bench), per-phase sums of the trace build
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r4e; mkdir -p $OUT; cd $R
export TMPDIR=/tmp
( timeout 300 python bench.py --no-cpu --no-extras --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err )
( timeout 300 python bench.py --no-cpu --no-extras --units 8192 --steps 10 --warmup 3 --exp > $OUT/bench8192.json 2>> $OUT/bench.err )
( timeout 300 python bench.py --no-cpu --no-extras --units 1024 --steps 20 --warmup 3 --exp > $OUT/bench1024.json 2>> $OUT/bench.err )
( MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace.so timeout 200 python tools/pipe_trace.py 4096 > $OUT/phases.txt 2>&1 )
for f in bench bench8192 bench1024; do python - <<PY
import json
try:
    j = json.loads(open("$OUT/$f.json").read().strip().splitlines()[-1])
    print("$f", j["ms_per_step"], j["value"], j.get("roofline", {}).get("frac"), j.get("bit_exact"))
except Exception as e: print("$f", "ERR", e)
PY
done; tail -3 $OUT/bench.err; head -18 $OUT/phases.txt
