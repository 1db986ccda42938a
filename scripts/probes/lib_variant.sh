#!/bin/bash
# one source file of the library with extra -D flags (timing experiments, results wrong) linked against the objects of the regular build:
#   bash scripts/probes/lib_variant.sh <tag> <file.hip> -DTN_ABL_NO_LOAD ...   ->  scripts/probes/bin/libpapr_<tag>.so   (use with PAPR_HIP_LIB=...)
set -e
tag=$1; src=$2; shift; shift
base=$(basename $src .hip)
mkdir -p scripts/probes/bin
extra=""; [ $base = chain4 ] && extra="-fno-slp-vectorize -mllvm -amdgpu-spill-vgpr-to-agpr=0"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -w $extra "$@" -c papr_amd/csrc/$base.hip -o scripts/probes/bin/${base}_$tag.o
objs=$(ls papr_amd/build/*.o | grep -v /$base.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs scripts/probes/bin/${base}_$tag.o -o scripts/probes/bin/libpapr_$tag.so -Wl,-rpath,/opt/rocm/lib
echo built scripts/probes/bin/libpapr_$tag.so
