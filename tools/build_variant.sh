#!/bin/bash
# analysis builds of the HIP library: tools/build_variant.sh <name> [-DFLAGS...] -> build/variants/libmspack_hip_<name>.so
# (run here, in the build container; the files travel to the GPU box; select one with MSPACK_HIP_SO)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -I include -c libmspack_amd/csrc/hip/shim.hip -o build/variants/shim_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libmspack_hip_$name.so build/variants/shim_$name.o libmspack_amd/csrc/host/*.o -lpthread
rm -f build/variants/shim_$name.o
echo build/variants/libmspack_hip_$name.so
