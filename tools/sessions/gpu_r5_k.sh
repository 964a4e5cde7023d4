# round 5, session k: the abort in test_copies_are_cut_at_pin_boundaries (final session): catch its message
mkdir -p gpurun_out/r5k; O=gpurun_out/r5k
n=0
for i in $(seq 1 30); do
  ( timeout 120 python -m pytest tests/test_gpu_hostpath.py -q -x -s -p no:cacheprovider -m gpu -k "headline_batch_host or copies_are_cut" ) > $O/run_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then n=$((n+1)); echo "run $i rc=$rc"; grep -v "^  File\|^$" $O/run_$i.log | head -30; fi
done
echo "failures: $n of 30"
