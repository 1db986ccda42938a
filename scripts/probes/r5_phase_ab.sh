# A/B of PAPR_C4_PHASE (thousand cycles the slack workgroups of a fused run start late): step time and per-launch times of the fused runs
mkdir -p gpurun_out/r5b
for PH in ${PHASES:-0 60 90 110 125 140 160 0}; do
  echo "=== PAPR_C4_PHASE=$PH"
  PAPR_C4_PHASE=$PH PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 2> gpurun_out/r5b/ph.err | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.3f  chain avg %.4f ms' % (j['ms_per_step'], j['roofline']['avg_launch_ms']))"
  grep "^kernel  9\|^kernel 10" gpurun_out/r5b/ph.err | head -8
done
