# LIBENV="lib:VAR=x,VAR2=y lib2: ..."  (lib = base or a variant tag)
mkdir -p gpurun_out/r5b
for LE in ${LIBENV}; do
  L=${LE%%:*}; E=${LE#*:}; E="${E//,/ }"
  if [ "$L" = base ]; then unset PAPR_HIP_LIB; else export PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_$L.so; fi
  echo "=== lib $L env $E"
  env $E PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 2> gpurun_out/r5b/le.err | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.3f' % (j['ms_per_step']))"
  grep "^kernel  9\|^kernel 10" gpurun_out/r5b/le.err | grep "M=512000"
done
