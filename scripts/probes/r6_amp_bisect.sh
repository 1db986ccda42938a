#!/bin/bash
# which of the round's late changes breaks long use_amp training (seed 1 collapsed after step 10,000): 15,000-step runs
O=gpurun_out/r6bis; mkdir -p $O; T=$(mktemp -d)
run() { name=$1; shift; ( time env "$@" python3 train.py --opt configs/nerfsyn/chair.yml --steps 15000 --set use_amp=true training.losses.lpips=0 seed=1 index=bis_$name save_dir=$T ) > $O/$name.log 2>&1
  echo "$name: $(grep 'Eval step' $O/$name.log | sed -n '20p' | cut -c1-90) | $(grep 'Eval step' $O/$name.log | tail -1 | cut -c1-100)" | tee -a $O/summary.txt; }
run tail0 PAPR_TAIL_F16=0
run tr0 PAPR_TN_TR=0
run both0 PAPR_TAIL_F16=0 PAPR_TN_TR=0
run both0_wb PAPR_TAIL_F16=0 PAPR_TN_TR=0 PAPR_LN_IN_FEATURES=0
rm -rf $T
