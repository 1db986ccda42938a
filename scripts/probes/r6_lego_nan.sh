#!/bin/bash
# the fp32 lego run (21,500-step schedule, seed 1): the first step whose gradients are not finite, its inputs and the model dumped
O=gpurun_out/r6lego; mkdir -p $O; T=$(mktemp -d)
( time env PYTHONUNBUFFERED=1 PAPR_DEBUG_NANCHECK_FROM=10000 PAPR_DEBUG_DUMP=$PWD/$O/nan_step.pt python3 train.py --opt configs/nerfsyn/lego.yml --steps 21500 --set use_amp=false training.losses.lpips=0 seed=1 index=nan_repro save_dir=$T eval.step=100000 ) > $O/nan_repro.log 2>&1
grep "non-finite\|Pruned" $O/nan_repro.log | tail -12; grep "Train step" $O/nan_repro.log | tail -2 | cut -c1-160; ls -la $O/nan_step.pt
rm -rf $T
