cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ic
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/ic/p -o pmc -- python $R/bench.py --steps 2 --warmup 1 --exp > $R/gpurun_out/ic/log.txt 2>&1
tail -3 $R/gpurun_out/ic/log.txt | cut -c1-300
python - <<'PY'
import csv, glob, os, collections
R=os.environ['GRAFT_REPO_ROOT']
for f in glob.glob(R+'/gpurun_out/ic/p/**/*counter_collection.csv', recursive=True):
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        if 'mspack_decode_lzx' in row.get('Kernel_Name',''):
            agg[row['Counter_Name']]+=float(row['Counter_Value']); n[row['Counter_Name']]+=1
    for k in agg: print('%-22s %.5g per dispatch (%d)'%(k,agg[k]/n[k],n[k]))
PY
