#!/bin/bash
# HBM-side traffic of the headline LZX launch: FETCH_SIZE / WRITE_SIZE in separate --pmc passes (kernel-trace only), summed
# over the launch's kernels (mspack_lzx_pipe_map + mspack_lzx_pipe + mspack_decode_lzx); writes gpurun_out/traffic/traffic.json
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/traffic; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
A="python $R/bench.py --exp --no-cpu --no-extras --steps 3 --warmup 1"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o pmc -- $A > $OUT/$c.log 2>&1
done
python - <<PY
import csv, glob, collections, json
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            if k in ("mspack_lzx_pipe_map", "mspack_lzx_pipe", "mspack_decode_lzx"):
                per[k] += float(row["Counter_Value"]); n[k] += 1
    tot[c] = {k: per[k] / n[k] for k in per}
    print(c, {k: round(v) for k, v in tot[c].items()}, "KiB per dispatch")
json.dump({"fetch_kib_per_launch": round(sum(tot["FETCH_SIZE"].values())), "write_kib_per_launch": round(sum(tot["WRITE_SIZE"].values())),
           "per_kernel_kib": tot}, open("$OUT/traffic.json", "w"), indent=1)
PY
cat $OUT/traffic.json
