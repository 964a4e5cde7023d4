#!/bin/bash
# scratch: build an experiment variant of the HIP library:  tools/mkvariant.sh name -DFLAG ...
n=$1; shift
cd /root/repo/libmspack_amd/csrc/hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -I../../../include "$@" -c shim.hip -o /tmp/shim_$n.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../exp_$n.so /tmp/shim_$n.o ../host/*.o -lpthread && echo built exp_$n.so
