# round 5, session j: the parse task's hot path peeled (mixed frames out of line) -- headline back?  + affected parity tests
mkdir -p gpurun_out/r5j; O=gpurun_out/r5j; R=$(pwd)
( timeout 600 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_lzx_frames.py tests/test_gpu_runs.py tests/test_gpu_large_files.py tests/test_gpu_kat.py tests/test_chm_extract.py tests/test_gpu_mszip_blocks.py -k "not launch_paths" ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2; do
  ( timeout 300 python bench.py --steps 30 --warmup 5 --no-extras --no-cpu ) > $O/bench_$i.json 2> $O/bench_$i.err
  python - <<P
import json
d = json.loads(open("$O/bench_$i.json").read().strip().splitlines()[-1])
print("run $i", d["ms_per_step"], d["roofline"]["achieved"], d["config"]["step_ms_min_median_max"])
P
done
( timeout 300 python tools/bench_folder_chain.py 4096 ) > $O/folder_chain.log 2>&1; grep -E "blocks" $O/folder_chain.log
