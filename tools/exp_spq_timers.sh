#!/bin/bash
# Analysis build (-DSPQ_TIMERS): where the match resolver's cycles go, for the wave of block 0, per kernel shape.
cd "$(dirname "$0")/.."
cp libmspack_amd/libmspack_hip.so /tmp/libmspack_hip.keep
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -DSPQ_TIMERS -I include \
  -c libmspack_amd/csrc/hip/shim.hip -o /tmp/shim_timers.o 2>/dev/null || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmspack_amd/libmspack_hip.so /tmp/shim_timers.o \
  libmspack_amd/csrc/host/*.o -lpthread
python - <<'PY'
import ctypes, subprocess, sys, os, json
sys.path.insert(0, '.')
import numpy as np
import libmspack_amd as M
L = M.lib()
L.mspack_hip_debug_counters.argtypes = [ctypes.c_void_p]
def read(tag, launches):
    a = np.zeros(8, dtype=np.uint64); L.mspack_hip_debug_counters(a.ctypes.data)
    g = max(int(a[4]), 1)
    print("%-34s per launch: cover %8d  load issue %7d  load wait %8d  stores %7d  tail %7d clk | %5d groups, %5d chunks; per group: cover %5d wait %5d" %
          (tag, a[0] / launches, a[1] / launches, a[2] / launches, a[3] / launches, a[6] / launches, a[4] / launches, a[5] / launches,
           a[0] / g, a[2] / g))
pass
# (a) headline batch, serial kernel; (b) with frame tables (unit kernel = commit only)
for ft in (False, True):
    plain, comp, off, ln, tab = M.corpus_lzx_units(0xBA5E11, 0, 4096, 65536, 21, frame_tables=True)
    units, out_bytes = M.make_units(M.KIND_LZX, off, ln + 4, np.full(4096, 65536), window_bits=21, reset_frames=2, frame_tabs=tab if ft else None)
    a = np.zeros(8, dtype=np.uint64); L.mspack_hip_debug_counters(a.ctypes.data)
    for _ in range(3):
        out, res = M.decode_batch(units, comp, out_bytes)
    assert (res["err"] == 0).all() and np.array_equal(out[:4096 * 65536], plain)
    read("LZX 4096 x 64 KiB, frame tables %s" % ft, 3)
PY
cp /tmp/libmspack_hip.keep libmspack_amd/libmspack_hip.so
