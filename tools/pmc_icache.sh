# instruction-cache counters of the headline launch's kernels (rocprofv3 --pmc, own pass; gpurun -- 'bash tools/pmc_icache.sh')
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ic; mkdir -p $R/gpurun_out/ic
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/ic/p -o pmc -- python $R/bench.py --steps 2 --warmup 1 --exp --no-cpu --no-extras > $R/gpurun_out/ic/log.txt 2>&1
rocprofv3 --pmc SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_INPUT_VALID_READY SQC_ICACHE_INPUT_VALID_READYB SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/ic/p2 -o pmc -- python $R/bench.py --steps 2 --warmup 1 --exp --no-cpu --no-extras > $R/gpurun_out/ic/log2.txt 2>&1
tail -2 $R/gpurun_out/ic/log.txt | cut -c1-200
python - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/ic/icache.txt
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
for f in sorted(glob.glob(R+'/gpurun_out/ic/p*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        k=row.get('Kernel_Name','')
        if 'mspack' in k:
            key=(k.split('(')[0], row['Counter_Name']); agg[key]+=float(row['Counter_Value']); n[key]+=1
    for k in sorted(agg): print('%-22s %-30s %.5g per dispatch (%d)'%(k[0],k[1],agg[k]/n[k],n[k]))
PY
