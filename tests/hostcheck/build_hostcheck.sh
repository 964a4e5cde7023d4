#!/bin/bash
# TEST INFRASTRUCTURE: tests/_build/hostcheck_{asan,tsan} = the HOST half of libmspack_amd/csrc/hip/shim.hip + the C host drivers +
# the CPU stand-in for a launch + the oracle + the corpus generators, one executable per sanitizer (tests/hostcheck/hostcheck_runtime.cpp
# says what is modelled).  usage: tests/hostcheck/build_hostcheck.sh asan|tsan
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
K=${1:-asan}
case $K in asan) SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined";; tsan) SAN="-fsanitize=thread";; *) echo "asan|tsan"; exit 2;; esac
SHIM=${HOSTCHECK_SHIM:-$R/libmspack_amd/csrc/hip/shim.hip}     # (experiments: another version of the file)
OUTX=${HOSTCHECK_OUT:-$R/tests/_build/hostcheck_$K}
B=$OUTX.d; mkdir -p $B
CXX=/opt/rocm/lib/llvm/bin/clang++; CC=/opt/rocm/lib/llvm/bin/clang
F="-O1 -g -fPIC -fno-omit-frame-pointer $SAN"
F0="-O0 -g -fPIC -fno-omit-frame-pointer $SAN"     # (shim.hip carries the kernels' sources too: unoptimised it compiles in 10 s instead of 50)
$CXX $F0 -std=c++17 -x c++ -DMSPACK_HOST_CHECK -Wno-unknown-attributes -Wno-unused-value -Wno-ignored-attributes \
  -I $R/tests/emu/include -I $R/include -I $R/libmspack_amd/csrc/hip -c $SHIM -o $B/shim.o
$CXX $F -std=c++17 -DMSPACK_HOST_CHECK -I $R/tests/emu/include -I $R/include -c $R/tests/hostcheck/hostcheck_runtime.cpp -o $B/runtime.o
$CXX $F -std=c++17 -I $R/include -c $R/tests/hostcheck/hostcheck_main.cpp -o $B/main.o
objs=""
for c in $R/libmspack_amd/csrc/host/*.c $R/oracle/*_oracle.c $R/libmspack_amd/csrc/corpus/*.c; do
  o=$B/$(basename ${c%.c}).o; $CC $F -Wall -Wno-unused-function -I $R/include -c $c -o $o; objs="$objs $o"
done
$CC $F -DSTANDIN_NO_ABI -Wno-comment -I $R/include -c $R/tests/csrc/batch_standin.c -o $B/standin.o
$CXX $SAN -o $OUTX $B/shim.o $B/runtime.o $B/main.o $B/standin.o $objs -lpthread -lm
echo built $OUTX
