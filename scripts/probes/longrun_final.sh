mkdir -p gpurun_out/long
for s in 1 2 3; do python3 train.py --opt configs/nerfsyn/chair.yml --steps 21500 --set use_amp=false training.losses.lpips=0 seed=$s index=abi18_$s save_dir=/tmp/papr_final 2>&1 | grep -E "^Eval step|Pruned|Added|^Train step: (5000|10000|15000|20000|21400)" | sed 's/ time: .*//' > gpurun_out/long/abi18_seed$s.log; rm -rf /tmp/papr_final; done
for s in 1 2 3; do echo seed $s; grep "^Eval step" gpurun_out/long/abi18_seed$s.log | tail -n 1; done
