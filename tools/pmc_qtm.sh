#!/bin/bash
# scratch: SQ counters + kernel stats for the Quantum kernel (tools/bench_codecs.py qtm); output gpurun_out/prof_qtm/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_qtm; rm -rf $OUT; mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python tools/bench_codecs.py qtm --iters 3 > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/p1 -o p1 -- python tools/bench_codecs.py qtm --iters 2 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $OUT/p2 -o p2 -- python tools/bench_codecs.py qtm --iters 2 > $OUT/p2.log 2>&1
python - <<'PY' | tee $OUT/summary.txt
import glob, csv, collections, os
R=os.environ['GRAFT_REPO_ROOT']
print(open(R+'/gpurun_out/prof_qtm/stats.log').read().strip().splitlines()[-1])
for f in glob.glob(R+'/gpurun_out/prof_qtm/stats/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'mspack' in r['Name']: print('kernel_stats: %s calls %s avg %.3f ms' % (r['Name'][:24], r['Calls'], float(r['AverageNs'])/1e6))
for f in sorted(glob.glob(R+'/gpurun_out/prof_qtm/p*/*counter_collection.csv')):
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        if 'mspack_decode_qtm' in row.get('Kernel_Name',''):
            agg[row['Counter_Name']]+=float(row['Counter_Value']); n[row['Counter_Name']]+=1
    for k in sorted(agg): print('  %-24s %.5g per dispatch' % (k, agg[k]/max(n[k],1)))
PY
