#!/bin/bash
# Analysis build: per-phase cycle counters of one wave's token commit (MSZIP), printed by block 0.
#   gpurun -- 'bash tools/exp_phase_timers.sh'
set -e
cd "$(dirname "$0")/.."
cp libmspack_amd/libmspack_hip.so /tmp/libmspack_hip.keep
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DZIP_PHASE_TIMERS $EXTRA -I include \
  -c libmspack_amd/csrc/hip/shim.hip -o /tmp/shim_timers.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmspack_amd/libmspack_hip.so /tmp/shim_timers.o \
  libmspack_amd/csrc/host/*.o -lpthread
python tools/bench_mszip_folder.py 512 8 2>&1 | grep -E "zip_run_tokens|spq_resolve|block_parse': True" | head -${LINES_:-8}
[ -n "$ONLY_FIRST" ] || python tools/bench_mszip_folder.py 2 64 2>&1 | grep -E "zip_run_tokens|spq_resolve|block_parse': True" | sort | uniq -c | sort -rn | head -8
cp /tmp/libmspack_hip.keep libmspack_amd/libmspack_hip.so
