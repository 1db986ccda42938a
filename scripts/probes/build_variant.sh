#!/bin/bash
# build the library from the current sources into scripts/probes/bin/libpapr_<tag>.so (extra -D flags after the tag)
# (chain2.hip is built without packed-fp32 VALU like papr_amd/build.py does: see EXTRA_FLAGS there; C2_SLP=1 builds it with)
set -e
tag=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function -w -Iinclude"
mkdir -p scripts/probes/bin/obj_$tag
objs=""
for f in papr_amd/csrc/*.hip; do
    o=scripts/probes/bin/obj_$tag/$(basename $f .hip).o
    X=""
    case "$(basename $f)" in chain2.hip) if [ "$C2_SLP" != "1" ]; then X=-fno-slp-vectorize; fi;; chain3.hip) X="-fno-slp-vectorize -mllvm -amdgpu-spill-vgpr-to-agpr=0";; esac
    /opt/rocm/bin/hipcc $FLAGS $X "$@" -c $f -o $o &
    objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o scripts/probes/bin/libpapr_$tag.so -Wl,-rpath,/opt/rocm/lib
echo built scripts/probes/bin/libpapr_$tag.so
