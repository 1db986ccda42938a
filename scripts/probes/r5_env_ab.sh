# A/B of environment settings on the bench step: SETS="VAR=a VAR=b ..." (each entry one run; "none" = default environment)
mkdir -p gpurun_out/r5b
for S in ${SETS}; do
  echo "=== $S"
  if [ "$S" = none ]; then E=""; else E="${S//,/ }"; fi   # (several variables: comma-separated)
  env $E PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 $BENCH_ARGS 2> gpurun_out/r5b/env.err | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.3f  chain avg %.4f ms  loss %.9f' % (j['ms_per_step'], j['roofline']['avg_launch_ms'], j['config']['final_loss']))"
  grep "^kernel  9\|^kernel 10" gpurun_out/r5b/env.err | grep "M=512000"
done
