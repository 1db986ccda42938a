#!/bin/bash
# build the library from the current sources into scripts/probes/bin/libpapr_<tag>.so (extra -D flags after the tag)
set -e
tag=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function -w -Iinclude"
mkdir -p scripts/probes/bin
# (chain2.hip is built without packed-fp32 VALU like papr_amd/build.py does: see EXTRA_FLAGS there; C2_SLP=1 builds it with)
OBJ=scripts/probes/bin/chain2_$tag.o
if [ "$C2_SLP" = "1" ]; then X=; else X=-fno-slp-vectorize; fi
/opt/rocm/bin/hipcc $FLAGS $X "$@" -c papr_amd/csrc/chain2.hip -o $OBJ
/opt/rocm/bin/hipcc $FLAGS "$@" -shared $(ls papr_amd/csrc/*.hip | grep -v chain2.hip) $OBJ -o scripts/probes/bin/libpapr_$tag.so -Wl,-rpath,/opt/rocm/lib
echo built scripts/probes/bin/libpapr_$tag.so
