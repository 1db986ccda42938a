#!/bin/bash
# the step with the weight gradients of the full 256 x 256 layers on gemm_tn_tr_kernel (PAPR_TN_TR=1) against the register-staged kernel, one box
O=gpurun_out/r6ts; mkdir -p $O
run() { PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 20 --warmup 5 --no-amp-line --no-shipped-line --psnr-steps 0 --no-cpu-baseline "$@" 2> $O/l.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(j['ms_per_step'],3), j['config']['final_loss'])"; grep "^kernel  *8 .*M=512000" $O/l.txt | head -4; }
{ for rep in 1 2; do
    echo "=== default, tr 0"; PAPR_TN_TR=0 run; echo "=== default, tr 1"; PAPR_TN_TR=1 run
    echo "=== amp, tr 0"; PAPR_TN_TR=0 run --amp; echo "=== amp, tr 1"; PAPR_TN_TR=1 run --amp
  done; } > $O/bench.txt 2>&1; cat $O/bench.txt
