#!/bin/bash
# build the library from the current sources into scripts/probes/bin/libpapr_<tag>.so (extra -D flags after the tag)
set -e
tag=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-function -w -Iinclude"
/opt/rocm/bin/hipcc $FLAGS "$@" -shared papr_amd/csrc/*.hip -o scripts/probes/bin/libpapr_$tag.so -Wl,-rpath,/opt/rocm/lib
echo built scripts/probes/bin/libpapr_$tag.so
