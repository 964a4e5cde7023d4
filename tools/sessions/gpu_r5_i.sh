# round 5, session i: where did the headline's 2-3 % go?  shipped (cheap run precheck) vs everything through the queue
mkdir -p gpurun_out/r5i; O=gpurun_out/r5i; R=$(pwd)
for v in shipped noruns shipped; do
  if [ $v = shipped ]; then SO=""; else SO="$R/build/variants/libmspack_hip_$v.so"; fi
  ( MSPACK_HIP_SO=$SO timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu ) > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<P
import json
d = json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1])
print("$v", d["ms_per_step"], d["roofline"]["achieved"], d["config"]["step_ms_min_median_max"])
P
done
( timeout 300 python tools/bench_folder_chain.py 4096 ) > $O/folder_chain.log 2>&1; grep -E "blocks" $O/folder_chain.log
