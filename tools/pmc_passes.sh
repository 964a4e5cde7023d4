#!/bin/bash
# scratch: PMC passes for the decode kernel (separate runs per counter group, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           ${PMC_MORE:+"SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC"} \
           ${PMC_MORE:+"GRBM_GUI_ACTIVE GRBM_COUNT"}; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/prof/p$i -o p$i -- python $R/bench.py --steps 2 --warmup 1 --exp --units ${UNITS:-4096} > $R/gpurun_out/prof/p$i.log 2>&1
done
python - <<'PY'
import glob, csv, collections, os
R=os.environ['GRAFT_REPO_ROOT']
for f in sorted(glob.glob(R+'/gpurun_out/prof/p*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(float); n=collections.defaultdict(int)
    for row in csv.DictReader(open(f)):
        if 'mspack_decode_lzx' in row.get('Kernel_Name',''):
            agg[row['Counter_Name']]+=float(row['Counter_Value']); n[row['Counter_Name']]+=1
    print(os.path.basename(f))
    for k in agg: print('  %-28s %.4g (per dispatch, %d dispatches)'%(k, agg[k]/max(n[k],1), n[k]))
PY
