#!/bin/bash
# round 6, GPU call 1: the f16-rows experiment of the parity mode (PAPR_MLP_H3_F16ROWS) + where the one-product hot statement's time goes
O=gpurun_out/r6b; mkdir -p $O
python3 -c "import torch; print(torch.cuda.get_device_name(0))" > $O/box.txt 2>&1
# 1. the new mode's forms against each other and against h3
timeout 900 python3 -m pytest tests/test_hip_chain_variants.py -q -m gpu -k "f16_rows" -x -s > $O/variants.txt 2>&1; echo "variants rc $?" >> $O/rc.txt
timeout 900 python3 -m pytest tests/test_hip_rccl.py tests/test_hip_kernels.py -q -m gpu -k "bench_gpus_2 or without_an_own_form or outside_the_own or small_unet or mse" -x > $O/modes.txt 2>&1; echo "new tests rc $?" >> $O/rc.txt
# 2. the step: default against the new mode, twice each (same box)
run() { PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 20 --warmup 5 --no-amp-line --no-shipped-line --psnr-steps 0 --no-cpu-baseline "$@" 2> $O/l.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(j['ms_per_step'],3), 'final_loss', j['config'].get('final_loss'))"; grep "^kernel  *\(9\|10\|8\) " $O/l.txt | head -12; }
for i in 1 2; do
  echo "=== h3"; run
  echo "=== h3_f16rows"; run --gemm-mode h3_f16rows
done > $O/bench_ab.txt 2>&1
echo "=== amp"; run --amp > $O/bench_amp.txt 2>&1
# 3. the goldens under the new mode (every bar unchanged)
PAPR_GEMM_MODE=h3_f16rows timeout 1500 python3 -m pytest tests/test_hip_model.py tests/test_dynamics_golden.py -q -m gpu > $O/goldens_f16rows.txt 2>&1; echo "goldens rc $?" >> $O/rc.txt
# 4. per-tensor gradient error against the reference goldens, both modes
for m in h3 h3_f16rows; do PAPR_GEMM_MODE=$m python3 scripts/probes/grad_err.py; done > $O/grad_err.txt 2>&1
# 5. the whole GPU suite on the tree as it stands
timeout 1700 python3 -m pytest tests -q -m gpu -x > $O/suite.txt 2>&1; echo "suite rc $?" >> $O/rc.txt
tail -3 $O/suite.txt
cat $O/rc.txt
