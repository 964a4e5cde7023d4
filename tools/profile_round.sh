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
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $R/bench.py $ARGS --no-cpu > $OUT/trace.log 2>&1
# separate PMC passes (never together with traces other than kernel-trace)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc_sq -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu > $OUT/pmc_sq.log 2>&1
python - <<'PY'
import csv, glob, os, collections, json
R = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()); OUT = R + '/gpurun_out/profile'
with open(OUT + '/summary.txt', 'w') as out:
    out.write(open(OUT + '/bench.json').read())
    for f in glob.glob(OUT + '/trace/**/*kernel_stats.csv', recursive=True):
        out.write('\n== rocprofv3 --kernel-trace --stats (%s)\n' % os.path.basename(f))
        for row in list(csv.reader(open(f)))[:8]: out.write(','.join(row) + '\n')
    for name in ('pmc_fetch', 'pmc_write', 'pmc_sq'):
        for f in glob.glob(OUT + '/%s/**/*counter_collection.csv' % name, recursive=True):
            agg = collections.defaultdict(float); n = collections.defaultdict(int)
            for row in csv.DictReader(open(f)):
                if 'mspack_decode' in row.get('Kernel_Name', ''):
                    agg[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
            out.write('\n== %s: per-dispatch averages for mspack_decode_* kernels\n' % name)
            for k in agg: out.write('%-24s %.6g  (dispatches %d)\n' % (k, agg[k] / max(n[k], 1), n[k]))
print(open(OUT + '/summary.txt').read())
PY
