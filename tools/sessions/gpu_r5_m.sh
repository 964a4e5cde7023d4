mkdir -p gpurun_out/r5m
( timeout 200 python tools/repro_pin_boundary2.py 10 ) > gpurun_out/r5m/out.log 2> gpurun_out/r5m/err.log; echo "rc=$?"
tail -14 gpurun_out/r5m/err.log
