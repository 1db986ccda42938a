#!/bin/bash
# instruction-cache counters of the fused-run kernel (is a pair of tiles' code -- 550 KB kernel -- resident in the 64 KB instruction cache two CUs share?)
export TMPDIR=/tmp
OUT=gpurun_out/icache; mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u > $OUT/avail.txt
for mode in inf keep; do
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $OUT/$mode -o ic -- python3 scripts/probes/chain_only.py $mode > /dev/null 2> $OUT/$mode.err
  echo "=== $mode"; python3 scripts/rocpd_pmc.py $(find $OUT/$mode -name "*.db") --keep mlp_chain --top 12
done > $OUT/icache.txt 2>&1
find $OUT -name "*.db" -delete
