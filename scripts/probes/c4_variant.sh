#!/bin/bash
# chain4.hip with extra -D flags (timing experiments, results wrong) linked against the objects of the regular build:
#   bash scripts/probes/c4_variant.sh <tag> -DC4_X_NOSTORE ...   ->  scripts/probes/bin/libpapr_<tag>.so   (use with PAPR_HIP_LIB=...)
set -e
tag=$1; shift
mkdir -p scripts/probes/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -w -fno-slp-vectorize -mllvm -amdgpu-spill-vgpr-to-agpr=0 "$@" -c papr_amd/csrc/chain4.hip -o scripts/probes/bin/chain4_$tag.o
objs=$(ls papr_amd/build/*.o | grep -v chain4.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs scripts/probes/bin/chain4_$tag.o -o scripts/probes/bin/libpapr_$tag.so -Wl,-rpath,/opt/rocm/lib
echo built scripts/probes/bin/libpapr_$tag.so
