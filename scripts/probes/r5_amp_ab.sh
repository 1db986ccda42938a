for S in ${SETS}; do
  echo "=== $S"
  if [ "$S" = none ]; then E=""; else E="${S//,/ }"; fi
  env $E python3 bench.py --steps 20 --warmup 5 --profile-steps 0 --no-cpu-baseline --no-amp-line --no-shipped-line --psnr-steps 0 --amp 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('amp ms_per_step %.3f loss %.9f' % (j['ms_per_step'], j['config']['final_loss']))"
done
