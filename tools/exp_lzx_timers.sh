#!/bin/bash
# Analysis build: cycle counters of the unit wave's token commit on the frame-parallel path, printed by a few blocks.
cd "$(dirname "$0")/.."
cp libmspack_amd/libmspack_hip.so /tmp/libmspack_hip.keep
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DLZX_PHASE_TIMERS $EXTRA -I include \
  -c libmspack_amd/csrc/hip/shim.hip -o /tmp/shim_timers.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmspack_amd/libmspack_hip.so /tmp/shim_timers.o \
  libmspack_amd/csrc/host/*.o -lpthread
python bench.py --exp --no-cpu --no-extras --steps 1 --warmup 0 $BENCH_ARGS 2>&1 | grep -E "lzx unit|lzx parse" | sort | uniq | head -${LINES_:-30}
cp /tmp/libmspack_hip.keep libmspack_amd/libmspack_hip.so
