#!/bin/bash
O=gpurun_out/r6trn; mkdir -p $O
{ timeout 1500 python3 -m pytest tests/test_hip_chain_variants.py -q -m gpu -x 2>&1 | tail -4
  timeout 900 python3 -m pytest tests/test_hip_dp.py tests/test_hip_h1.py -q -m gpu -x 2>&1 | tail -3; } > $O/tests.txt 2>&1
run() { PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 20 --warmup 5 --no-amp-line --no-shipped-line --psnr-steps 0 --no-cpu-baseline "$@" 2> $O/l.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(j['ms_per_step'],3), j['config']['final_loss'])"; grep "^kernel  *8 .*M=512000" $O/l.txt | head -5; }
{ echo "=== default"; run; echo "=== amp"; run --amp; } > $O/bench.txt 2>&1; cat $O/tests.txt $O/bench.txt
