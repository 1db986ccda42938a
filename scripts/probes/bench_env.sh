#!/bin/bash
# bench.py (10 steps, no extras) under each of the environment settings given as arguments ("NAME=VALUE" or "-" for none): step time + fused-run launch table
for e in "$@"; do
  if [ "$e" = "-" ]; then unset X; else export "$e"; fi
  PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 10 --warmup 3 --no-amp-line --no-shipped-line --psnr-steps 0 --no-cpu-baseline 2> /tmp/l.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$e]', round(j['ms_per_step'],3), 'ms/step')"
  grep "^kernel  *\(8\|9\|10\) " /tmp/l.txt | head -10
  if [ "$e" != "-" ]; then unset "${e%%=*}"; fi
done
