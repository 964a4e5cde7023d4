# round 5, session d: new kernels / paths on the hardware (checksum units, run fill, page-aligned arenas), then the numbers
mkdir -p gpurun_out/r5d; O=gpurun_out/r5d
( timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_runs.py tests/test_gpu_hostpath.py tests/test_cab_sticky.py tests/test_cabsets.py \
    tests/test_config2_cab.py tests/test_api_bench.py tests/test_gpu_large_files.py tests/test_gpu_mszip_blocks.py tests/test_gpu_messages.py tests/test_gpu_drivers.py \
    -k "not config5_shapes" ) > $O/pytest_affected.log 2>&1; echo "pytest affected rc=$?"; tail -4 $O/pytest_affected.log
( timeout 600 python tools/api_through.py 2 3 4 ) > $O/api_through.log 2>&1; echo "api_through rc=$?"; cat $O/api_through.log | tail -4
( MSPACK_TEST_LARGE=1 timeout 900 python tests/test_gpu_large_files.py ) > $O/large_files.log 2>&1; echo "large rc=$?"; tail -8 $O/large_files.log
( timeout 300 python tools/exp_bigfolder.py 512 ) > $O/bigfolder.log 2>&1; tail -8 $O/bigfolder.log
( timeout 300 python tools/bench_mszip_folder.py 2 2000 ) > $O/mszip_folder.log 2>&1; tail -6 $O/mszip_folder.log
( timeout 600 python bench.py ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'P'
import json
try:
    d = json.loads(open("gpurun_out/r5d/bench.json").read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "value_host_inclusive", "value_host_to_host")}, d["roofline"]["frac"])
    for s in d.get("secondary", []):
        print(s.get("config", "")[:60], s.get("ms_per_launch") or s.get("ms"), (s.get("through_api") or {}).get("MBps"))
except Exception as e:
    print("bench parse failed", e)
P
