# Long training runs on the final tree (train.py, the reference's driver protocol; procedural scene, MSE-only):
#   21,500 steps with pruning (from 10,000, every 500) and growth (20,000, 21,000) live, fp32 parity mode   -> train_21500_fp32.log
#   3,000 steps with the scene file's own use_amp: true                                                      -> train_3000_use_amp.log
OUT=gpurun_out/r5d; mkdir -p $OUT
T=$(mktemp -d)
( time python3 train.py --opt configs/nerfsyn/chair.yml --steps 3000 --set use_amp=true training.losses.lpips=0 seed=1 index=r5_amp save_dir=$T ) > $OUT/train_3000_use_amp.log 2>&1
grep "Eval step" $OUT/train_3000_use_amp.log | tail -3
( time python3 train.py --opt configs/nerfsyn/chair.yml --steps 21500 --set use_amp=false training.losses.lpips=0 seed=1 index=r5_fp32 save_dir=$T ) > $OUT/train_21500_fp32.log 2>&1
grep "Eval step\|prune\|add\|Prun\|Add" $OUT/train_21500_fp32.log | tail -12
rm -rf $T
