# the same 1,500-step training run under the library's A/B switches that claim bit-identical results: same logs?
mkdir -p gpurun_out/repro
run() { tag=$1; shift; env "$@" python3 train.py --opt configs/nerfsyn/chair.yml --steps 1500 --set use_amp=false training.losses.lpips=0 seed=1 index=sw_$tag save_dir=/tmp/papr_repro 2>&1 | grep -E "^Eval step|^Train step: [0-9]*00 " | sed 's/ time: .*//' > gpurun_out/repro/switch_$tag.log; rm -rf /tmp/papr_repro; }
run default PAPR_NOOP=1
run c4_two_role PAPR_C4_FUSED=0
run c4_generic PAPR_C4_GENERIC=1
run knn_t4 PAPR_KNN_T=4
for t in c4_two_role c4_generic knn_t4; do cmp -s gpurun_out/repro/switch_default.log gpurun_out/repro/switch_$t.log && echo "$t: IDENTICAL to the default run" || echo "$t: DIFFERS"; done
tail -1 gpurun_out/repro/switch_default.log
