#!/bin/bash
# round 6, session A (GPU box): WHO aborts in test_copies_are_cut_at_pin_boundaries?  Native backtrace of the aborting thread
# (tools/csrc/abort_bt.c, LD_PRELOAD; its log does not go through pytest's capture), the test's body 150x outside pytest, the test
# file as the suite runs it, the whole suite minus the five-minute shapes; then this round's starting bench line.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6a; mkdir -p $O
cd $R
gcc -O1 -g -shared -fPIC -o /tmp/abort_bt.so tools/csrc/abort_bt.c || exit 1
export LD_PRELOAD=/tmp/abort_bt.so
ABORT_BT_LOG=$O/abort_repro.txt timeout 600 python tools/repro_pin_boundary.py 150 > $O/repro1.log 2>&1; echo "repro1 rc=$?" >> $O/summary.txt
ABORT_BT_LOG=$O/abort_repro2.txt timeout 600 python tools/repro_pin_boundary2.py 12 > $O/repro2.log 2>&1; echo "repro2 rc=$?" >> $O/summary.txt
for i in 1 2 3; do
  ABORT_BT_LOG=$O/abort_file_$i.txt timeout 900 python -m pytest tests/test_gpu_hostpath.py -m gpu -x -q -s -k "not config5" > $O/hostpath_$i.log 2>&1
  echo "hostpath file run $i rc=$?" >> $O/summary.txt
done
for i in 1 2 3; do
  ABORT_BT_LOG=$O/abort_suite_$i.txt timeout 1200 python -m pytest tests -m gpu -x -q -s -k "not config5 and not large_files and not launch_paths_same_bytes" > $O/suite_$i.log 2>&1
  echo "suite run $i rc=$?" >> $O/summary.txt
  tail -3 $O/suite_$i.log >> $O/summary.txt
done
unset LD_PRELOAD
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
cat $O/summary.txt; cat $O/abort_*.txt 2>/dev/null | head -150
