#!/bin/bash
# chain4.hip, gemm.hip and rowops.hip with extra -D flags (h3_common.h constants: -DONE_TARGET_E_V=3 ...) -> scripts/probes/bin/libpapr_<tag>.so
set -e
tag=$1; shift
mkdir -p scripts/probes/bin
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -w"
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -mllvm -amdgpu-spill-vgpr-to-agpr=0 "$@" -c papr_amd/csrc/chain4.hip -o scripts/probes/bin/chain4_$tag.o &
/opt/rocm/bin/hipcc $F "$@" -c papr_amd/csrc/gemm.hip -o scripts/probes/bin/gemm_$tag.o &
/opt/rocm/bin/hipcc $F "$@" -c papr_amd/csrc/rowops.hip -o scripts/probes/bin/rowops_$tag.o &
wait
objs=$(ls papr_amd/build/*.o | grep -v "/chain4.o\|/gemm.o\|/rowops.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs scripts/probes/bin/chain4_$tag.o scripts/probes/bin/gemm_$tag.o scripts/probes/bin/rowops_$tag.o -o scripts/probes/bin/libpapr_$tag.so -Wl,-rpath,/opt/rocm/lib
echo built scripts/probes/bin/libpapr_$tag.so
