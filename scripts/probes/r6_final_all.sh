#!/bin/bash
# the round's final measurements on one box: bench set + timelines (r6_final_bench.sh), the profile sets of the default step and of use_amp, 21,500-step runs of both arms, seeds 1-3
bash scripts/probes/r6_final_bench.sh > gpurun_out/r6c_final.log 2>&1
bash scripts/profile_round.sh r06 > gpurun_out/r6c_profile.log 2>&1
bash scripts/profile_round.sh r06amp --amp > gpurun_out/r6c_profile_amp.log 2>&1
O=gpurun_out/r6c/seed; mkdir -p $O; rm -f $O/summary.txt; T=$(mktemp -d)
for arm in amp h3; do
  amp=false; [ $arm = amp ] && amp=true
  for seed in 1 2 3; do
    ( time python3 train.py --opt configs/nerfsyn/chair.yml --steps 21500 --set use_amp=$amp training.losses.lpips=0 seed=$seed index=r6f_${arm}_$seed save_dir=$T ) > $O/${arm}_seed$seed.log 2>&1
    echo "$arm seed $seed: $(grep 'Eval step' $O/${arm}_seed$seed.log | tail -1) | min GradScaler scale $(grep 'Train step' $O/${arm}_seed$seed.log | awk '{for(i=1;i<=NF;i++) if($i=="scale:") print $(i+1)}' | sort -g | head -1) | $(grep real $O/${arm}_seed$seed.log)" | tee -a $O/summary.txt
  done
done
rm -rf $T
tail -6 gpurun_out/r6c_final.log; cat $O/summary.txt
