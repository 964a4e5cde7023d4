cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof1; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/p1 -o p1 -- python $R/bench.py --steps 2 --warmup 1 --exp --no-cpu --no-extras > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/p2 -o p2 -- python $R/bench.py --steps 2 --warmup 1 --exp --no-cpu --no-extras > $OUT/p2.log 2>&1
python - <<'PY'
import glob, csv, collections, os
R=os.environ['GRAFT_REPO_ROOT']
for f in sorted(glob.glob(R+'/gpurun_out/prof1/p*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        k=row.get('Kernel_Name','')
        if 'mspack' in k:
            key=(k.split('(')[0], row['Counter_Name']); agg[key]+=float(row['Counter_Value']); n[key]+=1
    for k in sorted(agg): print('  %-24s %-26s %.5g'%(k[0], k[1], agg[k]/max(n[k],1)))
PY
