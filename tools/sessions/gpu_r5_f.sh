# round 5, session f: frames inside multi-frame blocks on the hardware; the per-folder chain by kernel; headline regression check
mkdir -p gpurun_out/r5f; O=gpurun_out/r5f; R=$(pwd)
( timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_lzx_frames.py tests/test_gpu_lzx.py tests/test_gpu_kat.py tests/test_gpu_runs.py \
    tests/test_gpu_large_files.py tests/test_gpu_lzx_log.py tests/test_gpu_fuzz.py tests/test_chm_extract.py tests/test_gpu_hostpath.py tests/test_gpu_qtm.py tests/test_cab_sticky.py \
    -k "not config5_shapes" ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r5f/prof_chain -- python $R/tools/bench_folder_chain.py 4096 ) > $R/$O/folder_chain.log 2>&1
cd $R; grep -E "blocks|adopted" $O/folder_chain.log
python - <<'P'
import csv, glob
for f in glob.glob("gpurun_out/r5f/prof_chain/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-28s calls %5s total %10.1f us avg %10.1f us" % (r["Name"][:28], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
P
( timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu ) > $O/bench_quick.json 2> $O/bench_quick.err; python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r5f/bench_quick.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["config"]["bit_exact"], d["config"]["units_on_frame_parallel_path"])
except Exception as e:
    print("bench parse failed", e)
P
