#!/bin/bash
# 21,500-step runs of lego.yml (LeakyReLU, a skip-layer value MLP: its training calls run in the parity arithmetic), use_amp false / true, seed 1
O=gpurun_out/r6lego; mkdir -p $O; rm -f $O/summary.txt; T=$(mktemp -d)
for amp in ${1:-false true}; do
  ( time env PYTHONUNBUFFERED=1 PAPR_DEBUG_NANCHECK_FROM=0 python3 train.py --opt configs/nerfsyn/lego.yml --steps 21500 --set use_amp=$amp training.losses.lpips=0 seed=1 index=lego_$amp save_dir=$T ) > $O/lego_amp_$amp.log 2>&1
  echo "lego use_amp=$amp seed 1: $(grep 'Eval step' $O/lego_amp_$amp.log | sed -n '20p' | cut -c1-90) | $(grep 'Eval step' $O/lego_amp_$amp.log | tail -1 | cut -c1-100) | min scale $(grep 'Train step' $O/lego_amp_$amp.log | awk '{for(i=1;i<=NF;i++) if($i=="scale:") print $(i+1)}' | sort -g | head -1) | points at the end $(grep 'Train step' $O/lego_amp_$amp.log | tail -1 | awk '{for(i=1;i<=NF;i++) if($i=="points:") print $(i+1)}') | faults $(grep -c 'Memory access fault' $O/lego_amp_$amp.log) | $(grep 'non-finite' $O/lego_amp_$amp.log | head -1 | cut -c1-120) | $(grep real $O/lego_amp_$amp.log)" | tee -a $O/summary.txt
done
rm -rf $T
