#!/bin/bash
# Seed spread of the long training run: 21,500 steps of chair.yml on the procedural scene with prune / add live,
# seeds 1..3, two arithmetic arms:
#   r1arith : PAPR_CHAIN=1 PAPR_OWN_ADAM=0 PAPR_UNET_REST=0  (round-1 kernels: chain.hip -- deleted later in round 3, this arm ran at commit 2b0e6d1 --, torch Adam, MIOpen for the U-Net's other layers)
#   default : this tree's kernels
# Run on the GPU box from the repo root:  bash scripts/seed_study.sh [steps] [seeds...]
# One log per run under gpurun_out/seed_study/ (Eval / Pruned / Added lines only), summary table at the end.
STEPS=${1:-21500}; shift
SEEDS=${@:-1 2 3}
OUT=gpurun_out/seed_study
mkdir -p $OUT
run() {   # arm seed env...
    arm=$1; seed=$2; shift 2
    env "$@" python3 train.py --opt configs/nerfsyn/chair.yml --steps $STEPS \
        --set use_amp=false training.losses.lpips=0 seed=$seed index=seed_${arm}_$seed save_dir=/tmp/papr_seed_study 2>&1 \
        | grep -E "^Eval step|Pruned|Added|^Train step: (5000|10000|15000|20000|21400)" > $OUT/${arm}_seed$seed.log
    rm -rf /tmp/papr_seed_study
}
for s in $SEEDS; do
    run r1arith $s PAPR_CHAIN=1 PAPR_OWN_ADAM=0 PAPR_UNET_REST=0
    run default $s PAPR_NOOP=1
done
echo "arm seed final_eval_psnr final_points" > $OUT/summary.txt
for f in $OUT/*_seed*.log; do
    b=$(basename $f .log)
    echo "$b $(grep '^Eval step' $f | tail -1 | awk '{print $NF}') $(grep '^Train step' $f | tail -1 | sed 's/.*points: \([0-9]*\).*/\1/')" >> $OUT/summary.txt
done
cat $OUT/summary.txt
