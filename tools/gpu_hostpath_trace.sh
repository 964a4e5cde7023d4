#!/bin/bash
# GPU box: kernel + memory-copy timeline of the host-buffer entry points for some chunk counts (rocprofv3 traces only, no counters)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for nc in ${CHUNKS:-2 8}; do
  rm -rf /tmp/hp_$nc
  EXP_HOSTPATH_WORKER=1 MSPACK_HIP_NCHUNKS=$nc rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/hp_$nc -o t -- python $R/tools/exp_hostpath.py 4096 1 > $R/gpurun_out/hp_trace_$nc.log 2>&1
  python - "$nc" <<'PY' > $R/gpurun_out/hp_timeline_$nc.txt 2>&1
import csv, glob, sys
nc = sys.argv[1]
ev = []
for f in glob.glob('/tmp/hp_%s/**/*kernel_trace.csv' % nc, recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:40], r.get('Queue_Id', '')))
for f in glob.glob('/tmp/hp_%s/**/*memory_copy_trace.csv' % nc, recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C ' + r.get('Direction', '') + ' ' + r.get('Bytes', r.get('Size', '')), ''))
ev.sort()
# the last to_device and the last to_host call: print the tail of the timeline
t0 = ev[0][0]
big = [e for e in ev]
for e in big[-140:]:
    print("%10.3f %10.3f  %8.3f ms  %s %s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[2], e[3]))
PY
done
