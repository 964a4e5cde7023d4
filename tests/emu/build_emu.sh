#!/bin/bash
# TEST INFRASTRUCTURE: tests/_build/libmspack_emu.so = the real kernel sources (libmspack_amd/csrc/hip/shim.hip) +
# the C host drivers, compiled for the host CPU on top of the wavefront emulator (tests/emu/emu_runtime.cpp).
# Never part of the product; tests select it with MSPACK_HIP_SO.   usage: tests/emu/build_emu.sh [extra -D flags]
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
B=$R/tests/_build; mkdir -p $B/emu
CXX=/opt/rocm/lib/llvm/bin/clang++
OUT=${EMU_OUT:-$B/libmspack_emu.so}
$CXX -O1 -g -std=c++17 -fPIC -fno-omit-frame-pointer -I $R/tests/emu/include -c $R/tests/emu/emu_runtime.cpp -o $B/emu/emu_runtime.o
$CXX ${EMU_OPT:--O0} -g -std=c++17 -fPIC -x c++ -fsanitize=thread -mllvm -tsan-instrument-func-entry-exit=false -fsanitize-coverage=bb,no-prune,trace-pc \
  -Wno-unknown-attributes -Wno-unused-value -Wno-ignored-attributes "$@" \
  -I $R/tests/emu/include -I $R/include -c $R/libmspack_amd/csrc/hip/shim.hip -o $B/emu/shim_emu.o
objs=""
for c in $R/libmspack_amd/csrc/host/*.c; do o=$B/emu/$(basename ${c%.c}).o; gcc -O2 -fPIC -Wall -I $R/include -c $c -o $o; objs="$objs $o"; done
$CXX -shared -fPIC -o $OUT $B/emu/shim_emu.o $B/emu/emu_runtime.o $objs -lpthread -ldl
# the back edges of the instrumented code: "<loop head> <back-edge source>" per line, module-relative
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $OUT 2>/dev/null | python3 -c '
import re, sys
for line in sys.stdin:
    m = re.match(r"^\s*([0-9a-f]+):\s+j[a-z]+\s+0x([0-9a-f]+)\s", line)
    if m and int(m.group(2), 16) <= int(m.group(1), 16):
        print(m.group(2), m.group(1))
' > $OUT.loops
echo built $OUT "($(wc -l < $OUT.loops) back edges)"
