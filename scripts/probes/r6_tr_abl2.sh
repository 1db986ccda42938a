#!/bin/bash
O=gpurun_out/r6tn; mkdir -p $O
{ timeout 900 python3 -m pytest tests/test_hip_chain_variants.py -q -m gpu -x -k "f16_rows_mode" 2>&1 | tail -3
  echo "== register kernel"; PAPR_GEMM_MODE=h1 python3 scripts/probes/chain_bench.py 2>&1 | tail -1
  for v in "" tr_NO_COMPUTE; do
    echo "== tr ${v:-as built}"
    if [ -n "$v" ]; then export PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_$v.so; else unset PAPR_HIP_LIB; fi
    PAPR_TN_TR=1 PAPR_GEMM_MODE=h1 python3 scripts/probes/chain_bench.py 2>&1 | tail -1
  done; } > $O/abl_tr4.txt 2>&1
cat $O/abl_tr4.txt
