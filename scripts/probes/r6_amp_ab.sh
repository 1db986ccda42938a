#!/bin/bash
O=gpurun_out/r6m; mkdir -p $O
t() { echo "=== $1"; shift; env "$@" python3 -m pytest tests/test_hip_amp_golden.py -q -m gpu -s -k "gradients and lego" 2>&1 | grep "^lego1k: rms\|w_k.bias " | cut -c1-250; }
{ t "prev lib, key+query ONE, value parity" PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_prev.so PAPR_AMP_MLP_ONLY=key,query
  t "prev lib, value ONE only" PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_prev.so PAPR_AMP_MLP_ONLY=value
  t "new lib, key ONE only" PAPR_AMP_MLP_ONLY=key
  t "new lib, query ONE only" PAPR_AMP_MLP_ONLY=query
  t "new lib, key+query" PAPR_AMP_MLP_ONLY=key,query
  t "prev lib, key only" PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_prev.so PAPR_AMP_MLP_ONLY=key
  t "prev lib, query only" PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_prev.so PAPR_AMP_MLP_ONLY=query
} > $O/ab.txt 2>&1
cat $O/ab.txt
