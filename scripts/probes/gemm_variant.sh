#!/bin/bash
# gemm.hip with extra -D flags (timing experiments, results may be wrong) linked against the objects of the regular build:
#   bash scripts/probes/gemm_variant.sh <tag> -DTN_EXP_X4 ...   ->  scripts/probes/bin/libpapr_<tag>.so   (use with PAPR_HIP_LIB=...)
set -e
tag=$1; shift
mkdir -p scripts/probes/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -w "$@" -c papr_amd/csrc/gemm.hip -o scripts/probes/bin/gemm_$tag.o
objs=$(ls papr_amd/build/*.o | grep -v "/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs scripts/probes/bin/gemm_$tag.o -o scripts/probes/bin/libpapr_$tag.so -Wl,-rpath,/opt/rocm/lib
echo built scripts/probes/bin/libpapr_$tag.so
