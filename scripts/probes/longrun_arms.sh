mkdir -p gpurun_out/long
run() { tag=$1; seed=$2; shift 2; env "$@" python3 train.py --opt configs/nerfsyn/chair.yml --steps 21500 --set use_amp=false training.losses.lpips=0 seed=$seed index=final_$tag save_dir=/tmp/papr_final 2>&1 | grep -E "^Eval step|Pruned|Added|^Train step: (5000|10000|15000|20000|21400)" > gpurun_out/long/$tag.log; rm -rf /tmp/papr_final; }
run seed1_knn_every_point 1 PAPR_KNN_BLOCKS=0
run seed1_scores_from_tail 1 PAPR_SCORES_IN_RUN=0
run final_kernels_seed3 3 PAPR_NOOP=1
for f in gpurun_out/long/*.log; do echo $f; grep "^Eval step" $f | tail -n 2; done
