#!/bin/bash
# bench.py (10 steps, no extras) on the regular library and on the variant libraries named on the command line: step time + per-shape launch table
run() { PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 10 --warmup 3 --no-amp-line --no-shipped-line --psnr-steps 0 --no-cpu-baseline 2> /tmp/l.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(j['ms_per_step'],3), 'ms/step')"; grep "^kernel  *\(9\|10\|8\) " /tmp/l.txt | head -8; }
run regular
for tag in "$@"; do export PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_$tag.so; run $tag; done
