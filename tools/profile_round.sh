#!/bin/bash
# Run ON THE GPU BOX (through gpurun): bench line + rocprofv3 kernel-trace stats + HBM traffic counters
# for the headline workload.  Outputs land in gpurun_out/profile/ ; copy the summaries to profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3"
python $R/bench.py $ARGS > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py $ARGS --no-cpu --no-extras > $OUT/trace.log 2>&1
# separate PMC passes (never together with traces other than kernel-trace)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extras > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extras > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc_sq -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extras > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extras > $OUT/pmc_sq2.log 2>&1
# the secondary configs (BASELINE 2 and 4): kernel stats of the same bench process, plus the frame-parallel LZX path
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_full -o trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu > $OUT/trace_full.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ft -o trace -- python $R/bench.py $ARGS --no-cpu --no-extras --frame-tables > $OUT/trace_ft.log 2>&1
python - <<'PY'
import csv, glob, os, collections, json
R = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()); OUT = R + '/gpurun_out/profile'
with open(OUT + '/summary.txt', 'w') as out:
    out.write(open(OUT + '/bench.json').read())
    for d, what in (('trace', 'headline run'), ('trace_full', 'bench.py with host_inclusive + secondary configs'), ('trace_ft', 'bench.py --frame-tables')):
        for f in glob.glob(OUT + '/%s/**/*kernel_stats.csv' % d, recursive=True):
            out.write('\n== rocprofv3 --kernel-trace --stats, %s (%s)\n' % (what, os.path.basename(f)))
            for row in list(csv.reader(open(f)))[:10]: out.write(','.join(row) + '\n')
    for name in ('pmc_fetch', 'pmc_write', 'pmc_sq', 'pmc_sq2'):
        for f in glob.glob(OUT + '/%s/**/*counter_collection.csv' % name, recursive=True):
            agg = collections.defaultdict(float); n = collections.defaultdict(int)
            for row in csv.DictReader(open(f)):
                if 'mspack_decode' in row.get('Kernel_Name', ''):
                    agg[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
            out.write('\n== %s: per-dispatch averages for mspack_decode_* kernels\n' % name)
            for k in agg: out.write('%-24s %.6g  (dispatches %d)\n' % (k, agg[k] / max(n[k], 1), n[k]))
print(open(OUT + '/summary.txt').read())
PY
