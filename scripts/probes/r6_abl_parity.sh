#!/bin/bash
O=gpurun_out/r6x; mkdir -p $O
run() { PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 20 --warmup 5 --no-amp-line --no-shipped-line --psnr-steps 0 --no-cpu-baseline "$@" 2> $O/l.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(j['ms_per_step'],3))"; grep "^kernel  *\(9\|10\) .*M=512000" $O/l.txt | head -4; }
{ echo "=== as built"; run; for ab in K KP1 KP2 NOST; do echo "=== abl_$ab"; PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_abl_$ab.so run; done; echo "=== as built"; run; } > $O/abl.txt 2>&1; cat $O/abl.txt
