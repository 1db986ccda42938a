#!/bin/bash
# the round's final measurements on one box: bench set + timelines (r6_final_bench.sh), the profile sets of the default step and of use_amp, one long run per arm
bash scripts/probes/r6_final_bench.sh > gpurun_out/r6c_final.log 2>&1
bash scripts/profile_round.sh r06 > gpurun_out/r6c_profile.log 2>&1
bash scripts/profile_round.sh r06amp --amp > gpurun_out/r6c_profile_amp.log 2>&1
O=gpurun_out/r6c/seed; mkdir -p $O; T=$(mktemp -d)
for arm in h3 amp; do
  amp=false; [ $arm = amp ] && amp=true
  ( time python3 train.py --opt configs/nerfsyn/chair.yml --steps 21500 --set use_amp=$amp training.losses.lpips=0 seed=1 index=r6f_${arm}_1 save_dir=$T ) > $O/${arm}_seed1.log 2>&1
  echo "$arm seed 1: $(grep 'Eval step' $O/${arm}_seed1.log | tail -1)  $(grep real $O/${arm}_seed1.log)" | tee -a $O/summary.txt
done
rm -rf $T
tail -6 gpurun_out/r6c_final.log; cat $O/summary.txt
