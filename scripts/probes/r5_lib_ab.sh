# A/B of variant libraries (scripts/probes/bin/libpapr_<tag>.so) on the bench step: LIBS="tag ..." (base = the regular library)
mkdir -p gpurun_out/r5b
for L in ${LIBS}; do
  if [ "$L" = base ]; then unset PAPR_HIP_LIB; else export PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_$L.so; fi
  echo "=== lib $L  $EXTRA"
  env $EXTRA PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 2> gpurun_out/r5b/lib.err | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.3f  chain avg %.4f ms  loss %.9f' % (j['ms_per_step'], j['roofline']['avg_launch_ms'], j['config']['final_loss']))"
  grep "^kernel  9\|^kernel 10" gpurun_out/r5b/lib.err | grep "M=512000"
  grep "^kernel  8" gpurun_out/r5b/lib.err | head -3
done
