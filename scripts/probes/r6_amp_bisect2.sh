#!/bin/bash
O=gpurun_out/r6bis; mkdir -p $O; T=$(mktemp -d)
run() { name=$1; shift; ( time env "$@" python3 train.py --opt configs/nerfsyn/chair.yml --steps 15000 --set use_amp=true training.losses.lpips=0 seed=1 index=bis_$name save_dir=$T ) > $O/$name.log 2>&1
  echo "$name: $(grep 'Eval step' $O/$name.log | sed -n '20p' | cut -c1-90) | $(grep 'Eval step' $O/$name.log | tail -1 | cut -c1-100)" | tee -a $O/summary.txt; }
run parity_mlps PAPR_H1_ROWS=f32
run e3 PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_e3.so
run e9 PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_e9.so
rm -rf $T
