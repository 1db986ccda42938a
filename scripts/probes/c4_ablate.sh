#!/bin/bash
# isolated 4-layer timings of chain4.hip and its timing-experiment variants (scripts/probes/c4_variant.sh)
echo "as built:"; python3 scripts/probes/chain_bench.py 2>/dev/null
for so in scripts/probes/bin/libpapr_*.so; do
    echo "$(basename $so .so):"; PAPR_HIP_LIB=$PWD/$so python3 scripts/probes/chain_bench.py 2>/dev/null
done
