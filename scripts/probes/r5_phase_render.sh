# the slack-workgroup delay (PAPR_C4_PHASE) on the 800 x 800 render and on lego / chair30k training
mkdir -p gpurun_out/r5b
for PH in ${PHASES:-0 6130 130 4130 0 6130}; do
  echo "=== PAPR_C4_PHASE=$PH"
  PAPR_C4_PHASE=$PH python3 scripts/bench_render.py 30000 400 2>&1 | grep render
  PAPR_C4_PHASE=$PH python3 bench.py --scene nerfsyn/lego.yml --points 30000 --steps 15 --warmup 4 --profile-steps 0 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lego30k ms_per_step %.3f' % j['ms_per_step'])"
  PAPR_C4_PHASE=$PH python3 bench.py --steps 15 --warmup 4 --profile-steps 0 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 --amp 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chair amp ms_per_step %.3f' % j['ms_per_step'])"
done
