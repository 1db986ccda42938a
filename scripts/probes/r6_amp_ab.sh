#!/bin/bash
O=gpurun_out/r6h; mkdir -p $O
python3 -m pytest tests/test_hip_amp_golden.py -q -m gpu -s -k "gradients and lego" > $O/new_lego.txt 2>&1
PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_prev.so python3 -m pytest tests/test_hip_amp_golden.py -q -m gpu -s -k "gradients and lego" > $O/prev_lego.txt 2>&1
PAPR_GEMM_MODE=h1 python3 scripts/probes/grad_err.py lego1k > $O/h1_new.txt 2>&1
PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_prev.so PAPR_GEMM_MODE=h1 python3 scripts/probes/grad_err.py lego1k > $O/h1_prev.txt 2>&1
echo done
