#!/bin/bash
# what binds gemm_tn_h3_kernel<., 2> (f16 x f16 rows): ablation builds (results wrong), the weight-gradient batch of a 4-layer 256-wide run in mode h1 (four FMT-2 jobs in one launch)
O=gpurun_out/r6tn; mkdir -p $O
{ for v in "" tn_NO_LOAD tn_NO_SPLIT tn_NO_COLSUM tn_NO_MFMA tn_LOADONLY; do
    echo "== ${v:-as built}"
    if [ -n "$v" ]; then export PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_$v.so; else unset PAPR_HIP_LIB; fi
    PAPR_GEMM_MODE=h1 python3 scripts/probes/chain_bench.py 2>&1 | tail -1
  done; } > $O/abl.txt 2>&1
cat $O/abl.txt
