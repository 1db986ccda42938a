mkdir -p gpurun_out/short
run() { tag=$1; shift; env "$@" python3 train.py --opt configs/nerfsyn/chair.yml --steps 6000 --set use_amp=false training.losses.lpips=0 seed=1 index=short_$tag save_dir=/tmp/papr_short 2>&1 | grep -E "^Eval step|^Train step: (1000|3000|5000|5900)" > gpurun_out/short/$tag.log; rm -rf /tmp/papr_short; }
run spatial_a PAPR_NOOP=1
run spatial_b PAPR_NOOP=1
run every_point_a PAPR_KNN_BLOCKS=0
run every_point_b PAPR_KNN_BLOCKS=0
for f in gpurun_out/short/*.log; do echo $f; grep "^Eval step" $f | awk '{printf "%s %s  ", $3, substr($NF,1,8)} END {print ""}'; grep "^Train step" $f | awk '{printf "%s %s  ", $3, substr($5,1,12)} END {print ""}'; done
