#!/bin/bash
# scratch: PMC passes for the LZX kernels (separate runs per counter group, as the guide prescribes: --pmc with
# --kernel-trace only).  Output: gpurun_out/prof/pmc_summary.txt (per kernel, per dispatch averages).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof
mkdir -p $OUT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           ${PMC_MORE:+"FETCH_SIZE"} ${PMC_MORE:+"WRITE_SIZE"}; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python $R/bench.py --steps 2 --warmup 1 --exp --no-cpu --no-extras --units ${UNITS:-4096} ${BENCH_ARGS} > $OUT/p$i.log 2>&1
done
python - <<'PY' | tee $OUT/pmc_summary.txt
import glob, csv, collections, os
R=os.environ['GRAFT_REPO_ROOT']
for f in sorted(glob.glob(R+'/gpurun_out/prof/p*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        k=row.get('Kernel_Name','')
        if 'mspack' in k:
            key=(k.split('(')[0], row['Counter_Name'])
            agg[key]+=float(row['Counter_Value']); n[key]+=1
    print(os.path.basename(f))
    for k in sorted(agg): print('  %-24s %-26s %.5g (per dispatch, %d dispatches)'%(k[0], k[1], agg[k]/max(n[k],1), n[k]))
PY
