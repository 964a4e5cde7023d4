#!/bin/bash
# scratch (GPU box): calibrate FETCH_SIZE / WRITE_SIZE on a known byte count in this kernel's own access
# patterns, as MI355X_MICROARCH.md asks for widths other than 16 B/lane: an incompressible corpus makes
# the LZX encoder emit stored blocks, which the kernel copies input -> output with 1-byte-per-lane loads
# (known: bytes read == bytes written == 4096 x 64 KiB).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/calib
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o pmc -- python $R/bench.py --text ${TEXT:-4} --steps 2 --warmup 1 --exp > $OUT/$c.log 2>&1
done
python - <<'PY'
import csv, glob, os, collections, json
R = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()); OUT = R + '/gpurun_out/calib'
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    print(open(OUT + '/%s.log' % c).read().strip().splitlines()[-1][:300])
    for f in glob.glob(OUT + '/%s/**/*counter_collection.csv' % c, recursive=True):
        agg = collections.defaultdict(float); n = collections.defaultdict(int)
        for row in csv.DictReader(open(f)):
            if 'mspack_decode_lzx' in row.get('Kernel_Name', ''):
                agg[row['Counter_Name']] += float(row['Counter_Value']); n[row['Counter_Name']] += 1
        for k in agg: print('%-12s %.6g KiB per dispatch (%d dispatches)' % (k, agg[k] / max(n[k], 1), n[k]))
PY
