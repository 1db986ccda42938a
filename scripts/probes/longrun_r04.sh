#!/bin/bash
# round 4's tree, 21,500 steps of the procedural chair scene (prune / add live), seeds 1-3: the quality check behind profiles/r04_long_runs/
mkdir -p gpurun_out/long4
for s in 1 2 3; do python3 train.py --opt configs/nerfsyn/chair.yml --steps 21500 --set use_amp=false training.losses.lpips=0 seed=$s index=r04_$s save_dir=/tmp/papr_r04 2>&1 | grep -E "^Eval step|Pruned|Added|^Train step: (5000|10000|15000|20000|21400)" | sed 's/ time: .*//' > gpurun_out/long4/r04_seed$s.log; rm -rf /tmp/papr_r04; done
for s in 1 2 3; do echo seed $s; grep "^Eval step" gpurun_out/long4/r04_seed$s.log | tail -n 1; done > gpurun_out/long4/summary.txt
