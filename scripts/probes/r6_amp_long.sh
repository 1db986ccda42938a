#!/bin/bash
# long use_amp runs (21,500 steps, seed in $1, default 1) with the weight gradients on gemm_tn_tr_kernel (default) and on the register-staged kernel
O=gpurun_out/r6long; mkdir -p $O; T=$(mktemp -d); seed=${1:-1}
run() { name=$1; shift; ( time env "$@" python3 train.py --opt configs/nerfsyn/chair.yml --steps 21500 --set use_amp=true training.losses.lpips=0 seed=$seed index=long_$name save_dir=$T ) > $O/$name.log 2>&1
  echo "$name seed $seed: $(grep 'Eval step' $O/$name.log | sed -n '20p' | cut -c1-90) | $(grep 'Eval step' $O/$name.log | tail -1 | cut -c1-100) | min scale $(grep 'Train step' $O/$name.log | awk '{for(i=1;i<=NF;i++) if($i=="scale:") print $(i+1)}' | sort -g | head -1)" | tee -a $O/summary.txt; }
run amp_default X=1
run amp_tr0 PAPR_TN_TR=0
rm -rf $T
