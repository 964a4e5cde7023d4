#!/bin/bash
# memory-side counters of the LZX kernels (one --pmc group per run)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmcmem; rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_WRITE_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TCP_TA_DATA_STALL_CYCLES_sum" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python $R/bench.py --steps 2 --warmup 1 --exp --no-cpu --no-extras ${BENCH_ARGS} > $OUT/p$i.log 2>&1 || tail -3 $OUT/p$i.log
done
python - <<'PY'
import glob, csv, collections, os
R=os.environ['GRAFT_REPO_ROOT']
for f in sorted(glob.glob(R+'/gpurun_out/pmcmem/p*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        k=row.get('Kernel_Name','')
        if 'mspack' in k:
            key=(k.split('(')[0], row['Counter_Name']); agg[key]+=float(row['Counter_Value']); n[key]+=1
    for k in sorted(agg): print('  %-22s %-32s %.5g' % (k[0], k[1], agg[k]/max(n[k],1)))
PY
