#!/bin/bash
B=$PWD/scripts/probes/bin
for tag in "$@"; do for mode in inf keep; do
  echo "=== $tag mode=$mode: per wave, slots 7 and 8 (P1 | barrier | A | B | C | D | end)"
  PAPR_HIP_LIB=$B/libpapr_$tag.so STAMPS=7 S0=7 S1=9 python3 scripts/probes/chain4_trace.py $mode 2>&1 | grep -v amdgpu.ids
done; done
