# round 5, session e: Quantum variants, the single-folder chain traced, the chunk policy / checksum fixes on the hardware
mkdir -p gpurun_out/r5e; O=gpurun_out/r5e; R=$(pwd)
( timeout 300 python -m pytest -x -q -m gpu -p no:cacheprovider tests/test_cabsets.py tests/test_cab_sticky.py tests/test_gpu_qtm.py tests/test_gpu_hostpath.py -k "not config5_shapes and not headline" ) > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
( timeout 300 python tools/bench_qtm_config4.py 128 32 ) > $O/qtm_shipped.log 2>&1; tail -2 $O/qtm_shipped.log
( MSPACK_HIP_SO=$R/build/variants/libmspack_hip_q_nosplit.so timeout 300 python tools/bench_qtm_config4.py 128 32 ) > $O/qtm_nosplit.log 2>&1; tail -2 $O/qtm_nosplit.log
( MSPACK_HIP_NCHUNKS=1 MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace.so timeout 300 python tools/pipe_trace_folder.py large 4096 ) > $O/trace_large.log 2>&1; tail -22 $O/trace_large.log
( MSPACK_HIP_NCHUNKS=1 MSPACK_HIP_SO=$R/build/variants/libmspack_hip_trace.so timeout 300 python tools/pipe_trace_folder.py text 512 ) > $O/trace_text.log 2>&1; tail -22 $O/trace_text.log
( timeout 600 python tools/api_through.py 2 4 ) > $O/api_through.log 2>&1; tail -3 $O/api_through.log
( MSPACK_HIP_TRACE=1 MSPACK_TEST_LARGE=1 timeout 900 python tests/test_gpu_large_files.py ) > $O/large_files.log 2>&1; grep -E "mspack_hip\[|extract\(\)" $O/large_files.log | tail -12
