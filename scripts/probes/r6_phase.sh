#!/bin/bash
# the late start of the workgroups with slack (PAPR_SW_C4_PHASE = 1000 x fewest steps + hundreds of cycles per step): re-tuned for f16 rows?
O=gpurun_out/r6q; mkdir -p $O
run() { PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 20 --warmup 5 --no-amp-line --no-shipped-line --psnr-steps 0 --no-cpu-baseline "$@" 2> $O/l.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(j['ms_per_step'],3))"; grep "^kernel  *\(9\|10\) .*M=512000" $O/l.txt | head -4; }
for ph in 6130 0 4130 6090 6170 4090 4170 6060; do echo "=== PAPR_C4_PHASE=$ph"; PAPR_C4_PHASE=$ph run; done > $O/phase.txt 2>&1
echo "=== again 6130"; PAPR_C4_PHASE=6130 run >> $O/phase.txt 2>&1
for ph in 0 4130 8130; do echo "=== amp PAPR_C4_PHASE=$ph"; PAPR_C4_PHASE=$ph run --amp; done >> $O/phase.txt 2>&1
grep "===\|ms_per" $O/phase.txt
