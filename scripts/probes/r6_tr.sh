#!/bin/bash
O=gpurun_out/r6y; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_amp_golden.py tests/test_hip_h1.py tests/test_hip_model.py -q -m gpu -k "amp or h1" > $O/t.txt 2>&1; tail -3 $O/t.txt | cut -c1-200
run() { PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 20 --warmup 5 --no-amp-line --no-shipped-line --psnr-steps 0 --no-cpu-baseline "$@" 2> $O/l.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(j['ms_per_step'],3), j['config']['final_loss'])"; grep "^kernel  *9 .*M=512000" $O/l.txt | head -2; }
{ echo "=== amp"; run --amp; echo "=== amp, write-back (PAPR_LN_IN_FEATURES=0 keeps it)"; PAPR_LN_IN_FEATURES=0 run --amp; echo "=== amp"; run --amp; } > $O/bench.txt 2>&1; cat $O/bench.txt
