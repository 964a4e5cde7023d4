mkdir -p gpurun_out/r5b; O=gpurun_out/r5b
( timeout 200 python tools/stress_chm_batch.py 90 ) > $O/stress_plain.log 2>&1; echo "stress_plain rc=$?"
tail -15 $O/stress_plain.log
( timeout 200 python tools/stress_chm_batch.py 60 mix ) > $O/stress_mix.log 2>&1; echo "stress_mix rc=$?"
tail -15 $O/stress_mix.log
for i in 1 2 3 4 5 6; do
  ( timeout 120 python -m pytest tests/test_chm_extract.py -m gpu -x -q -p no:cacheprovider ) > $O/chm_extract_$i.log 2>&1; echo "chm_extract $i rc=$?"
  grep -E "passed|failed|AssertionError" $O/chm_extract_$i.log | head -5
done
