mkdir -p gpurun_out/r5a; O=gpurun_out/r5a
run() { name=$1; shift; ( timeout 300 env "$@" ) > $O/$name.log 2>&1; echo "$name rc=$?" >> $O/summary.txt; }
run units_default python tools/repro_chm_lifetimes.py units 4
run driver_default python tools/repro_chm_lifetimes.py driver 4
run driver_pinout0 MSPACK_HIP_PIN_OUT=0 python tools/repro_chm_lifetimes.py driver 3
run driver_nchunks1 MSPACK_HIP_NCHUNKS=1 python tools/repro_chm_lifetimes.py driver 3
run units_nchunks1 MSPACK_HIP_NCHUNKS=1 python tools/repro_chm_lifetimes.py units 3
run units_noframes MSPACK_HIP_NO_FRAME_PARSE=1 python tools/repro_chm_lifetimes.py units 3
run units_ncompute1 MSPACK_HIP_NCOMPUTE=1 python tools/repro_chm_lifetimes.py units 3
cat $O/summary.txt
for f in $O/*.log; do echo "== $f"; tail -8 $f; done
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $O/pytest_full.log 2>&1; echo "pytest rc=$?"
tail -40 $O/pytest_full.log
