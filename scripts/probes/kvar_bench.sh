#!/bin/bash
# chain_bench.py on the regular library and on every variant library named on the command line (scripts/probes/bin/libpapr_<tag>.so)
python3 scripts/probes/chain_bench.py 2>&1 | tail -1 | sed 's/^/regular: /'
for tag in "$@"; do
    PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_$tag.so python3 scripts/probes/chain_bench.py 2>&1 | tail -1 | sed "s/^/$tag: /"
done
