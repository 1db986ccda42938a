for PH in 0 6050 6070 6090 6110 0; do
  echo "=== PAPR_C4_PHASE=$PH"
  PAPR_C4_PHASE=$PH PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 --amp 2>gpurun_out/r5b/ph2.err | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chair amp ms_per_step %.3f' % j['ms_per_step'])"
  grep "^kernel  9\|^kernel 10" gpurun_out/r5b/ph2.err | grep "M=512000"
done
