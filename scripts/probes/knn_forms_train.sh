# the same training run with the two forms of ray_knn: same bits?  (exact ties at the k-th distance are the only place where the forms may pick different points)
STEPS=${1:-3000}
mkdir -p gpurun_out/repro
run() { tag=$1; shift; env "$@" python3 train.py --opt configs/nerfsyn/chair.yml --steps $STEPS --set use_amp=false training.losses.lpips=0 seed=1 index=kf_$tag save_dir=/tmp/papr_repro 2>&1 | grep -E "^Eval step|Pruned|Added|^Train step: [0-9]*00 " | sed 's/ time: .*//' > gpurun_out/repro/knnform_$tag.log; rm -rf /tmp/papr_repro; }
run spatial PAPR_NOOP=1
run every_point PAPR_KNN_BLOCKS=0
cmp gpurun_out/repro/knnform_spatial.log gpurun_out/repro/knnform_every_point.log && echo "IDENTICAL logs over $STEPS steps" || { echo "logs differ"; diff gpurun_out/repro/knnform_spatial.log gpurun_out/repro/knnform_every_point.log | head -6; }
