# round 5, session g: staging pool, run segments, the drivers through it
mkdir -p gpurun_out/r5g; O=gpurun_out/r5g; R=$(pwd)
( timeout 900 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_gpu_hostpath.py tests/test_gpu_runs.py tests/test_gpu_mszip_blocks.py tests/test_gpu_mszip.py \
    tests/test_cab_sticky.py tests/test_cabsets.py tests/test_chm_extract.py tests/test_config2_cab.py tests/test_api_bench.py tests/test_gpu_large_files.py tests/test_gpu_drivers.py \
    tests/test_gpu_reference_suites.py tests/test_oab.py tests/test_szdd_kwaj.py -k "not config5_shapes" ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
( timeout 600 python tools/api_through.py 2 3 4 ) > $O/api_through.log 2>&1; tail -4 $O/api_through.log
( timeout 600 python tools/bench_folder_chain.py 4096 ) > $O/folder_chain.log 2>&1; grep -E "blocks" $O/folder_chain.log
( MSPACK_TEST_LARGE=1 timeout 900 python tests/test_gpu_large_files.py ) > $O/large_files.log 2>&1; grep -E "extract\(\)|match" $O/large_files.log | tail -5
