mkdir -p gpurun_out/r5c; O=gpurun_out/r5c
fails=0
for i in 1 2 3 4 5 6 7 8 9 10; do
  ( timeout 120 python -m pytest tests/test_chm_extract.py -m gpu -x -q -p no:cacheprovider ) > $O/chm_extract_$i.log 2>&1 || fails=$((fails+1))
done
echo "chm_extract: $fails of 10 runs failed"; grep -h -E "AssertionError" $O/chm_extract_*.log | head -3
( timeout 600 python -m pytest tests/test_gpu_hostpath.py -m gpu -x -q -p no:cacheprovider -k "many_decompressors or cut_at_pin or chunked or mixed or to_device" ) > $O/hostpath_new.log 2>&1; echo "hostpath_new rc=$?"; tail -5 $O/hostpath_new.log
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=8 > $O/pytest_full.log 2>&1; echo "pytest rc=$?"
tail -16 $O/pytest_full.log
