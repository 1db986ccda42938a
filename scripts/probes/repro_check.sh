# two training runs of the same configuration: are they the same bits?  usage: repro_check.sh [steps]   (eval PSNR and losses are printed with full precision;
# the wall time field is cut off before the comparison)
STEPS=${1:-2000}
mkdir -p gpurun_out/repro
for t in a b; do python3 train.py --opt configs/nerfsyn/chair.yml --steps $STEPS --set use_amp=false training.losses.lpips=0 seed=1 index=repro_$t save_dir=/tmp/papr_repro 2>&1 | grep -E "^Eval step|Pruned|Added|^Train step: [0-9]*00 " | sed 's/ time: .*//' > gpurun_out/repro/${STEPS}_$t.log; rm -rf /tmp/papr_repro; done
cmp gpurun_out/repro/${STEPS}_a.log gpurun_out/repro/${STEPS}_b.log && echo "IDENTICAL logs over $STEPS steps ($(wc -l < gpurun_out/repro/${STEPS}_a.log) lines)" || { echo "logs differ"; diff gpurun_out/repro/${STEPS}_a.log gpurun_out/repro/${STEPS}_b.log | head -8; }
tail -2 gpurun_out/repro/${STEPS}_a.log
