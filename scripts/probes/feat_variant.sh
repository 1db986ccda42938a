#!/bin/bash
# features.hip with extra -D flags (timing experiments, results wrong) linked against the objects of the regular build
set -e
tag=$1; shift
mkdir -p scripts/probes/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -w "$@" -c papr_amd/csrc/features.hip -o scripts/probes/bin/features_$tag.o
objs=$(ls papr_amd/build/*.o | grep -v "/features.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs scripts/probes/bin/features_$tag.o -o scripts/probes/bin/libpapr_$tag.so -Wl,-rpath,/opt/rocm/lib
echo built scripts/probes/bin/libpapr_$tag.so
