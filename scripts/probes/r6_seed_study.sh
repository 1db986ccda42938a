#!/bin/bash
# round 6: 21,500-step train.py runs (prune / add live, procedural scene, MSE-only), seeds 1-3:
#   arm h3          use_amp=false, default parity mode
#   arm h3_f16rows  use_amp=false, PAPR_GEMM_MODE=h3_f16rows (f16 rows kept for the weight gradients)
#   arm amp         use_amp=true (the scene file's own setting)
# usage: r6_seed_study.sh <outdir> <arm> [<arm> ...]
OUT=$1; shift; mkdir -p $OUT
T=$(mktemp -d)
for arm in "$@"; do
  for seed in 1 2 3; do
    case $arm in
      h3) env="PAPR_GEMM_MODE=h3"; amp=false;;
      h3_f16rows) env="PAPR_GEMM_MODE=h3_f16rows"; amp=false;;
      amp) env="PAPR_GEMM_MODE=h3"; amp=true;;
    esac
    ( time env $env python3 train.py --opt configs/nerfsyn/chair.yml --steps 21500 --set use_amp=$amp training.losses.lpips=0 seed=$seed index=r6_${arm}_$seed save_dir=$T ) > $OUT/${arm}_seed$seed.log 2>&1
    echo "$arm seed $seed: $(grep 'Eval step' $OUT/${arm}_seed$seed.log | tail -1)  $(grep real $OUT/${arm}_seed$seed.log)" | tee -a $OUT/summary.txt
  done
done
rm -rf $T
