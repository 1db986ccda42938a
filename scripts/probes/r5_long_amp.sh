# 21,500 steps of train.py with the scene file's own use_amp: true (prune / add live), final tree
OUT=gpurun_out/r5d; mkdir -p $OUT
T=$(mktemp -d)
( time python3 train.py --opt configs/nerfsyn/chair.yml --steps 21500 --set use_amp=true training.losses.lpips=0 seed=1 index=r5_amp_long save_dir=$T ) > $OUT/train_21500_use_amp.log 2>&1
grep "Eval step" $OUT/train_21500_use_amp.log | tail -4; grep "real" $OUT/train_21500_use_amp.log; grep "scale:" $OUT/train_21500_use_amp.log | tail -1 | cut -c1-200
rm -rf $T
