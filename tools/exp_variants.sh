#!/bin/bash
# Analysis builds: the same bench with parts of the LZX commit compiled out (results are then wrong on purpose:
# --exp skips the comparison) -- what each part costs in the unit kernel.   gpurun -- 'bash tools/exp_variants.sh'
# VARIANTS="base:-DX nocopy:-DLZX_EXP_NOCOPY ..." BENCH_ARGS="--frame-tables"
cd "$(dirname "$0")/.."
R=$PWD
cp libmspack_amd/libmspack_hip.so /tmp/libmspack_hip.keep
export TMPDIR=/tmp
for v in ${VARIANTS:-base: nocopy:-DLZX_EXP_NOCOPY nomatch:-DLZX_EXP_NOMATCH nolit:-DLZX_EXP_NOLIT}; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value ${flags//,/ } -I include \
    -c libmspack_amd/csrc/hip/shim.hip -o /tmp/shim_var.o 2>/dev/null || { echo "$name: build failed"; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmspack_amd/libmspack_hip.so /tmp/shim_var.o \
    libmspack_amd/csrc/host/*.o -lpthread
  rm -rf /tmp/prof_$name
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o v -- \
     python $R/bench.py --exp --no-cpu --no-extras --steps 6 --warmup 2 $BENCH_ARGS > /tmp/prof_$name.log 2>&1)
  echo "== $name ($flags)"
  python - "$name" <<'PY'
import csv, glob, sys
for f in glob.glob('/tmp/prof_%s/**/*kernel_stats.csv' % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        if 'mspack' in r['Name']:
            print('   %-24s calls %3s avg %9.3f ms' % (r['Name'].split('(')[0], r['Calls'], float(r['AverageNs']) / 1e6))
PY
done
cp /tmp/libmspack_hip.keep libmspack_amd/libmspack_hip.so
