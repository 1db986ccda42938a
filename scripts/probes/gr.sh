#!/bin/bash
# gpurun helper: bash scripts/probes/gr.sh <out-name> <command ...>  -> runs the command, output in gpurun_out/r5b/<out-name>.txt (printed as well)
mkdir -p gpurun_out/r5b
out=gpurun_out/r5b/$1.txt; shift
"$@" > $out 2>&1
cat $out
