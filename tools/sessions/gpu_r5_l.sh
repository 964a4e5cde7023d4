mkdir -p gpurun_out/r5l
( timeout 150 python tools/repro_pin_boundary.py 12 ) > gpurun_out/r5l/out.log 2> gpurun_out/r5l/err.log; echo "rc=$?"
tail -12 gpurun_out/r5l/err.log
