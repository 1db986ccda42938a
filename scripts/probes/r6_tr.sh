#!/bin/bash
O=gpurun_out/r6t; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_chain_variants.py -q -m gpu -k "f16_rows or one_product" -x > $O/variants.txt 2>&1; tail -15 $O/variants.txt | cut -c1-300
{ echo "tr:"; python3 scripts/probes/chain_bench.py 2>/dev/null; echo "no tr:"; PAPR_TN_TR=0 python3 scripts/probes/chain_bench.py 2>/dev/null; echo "h1 tr:"; PAPR_GEMM_MODE=h1 python3 scripts/probes/chain_bench.py 2>/dev/null; echo "h1 no tr:"; PAPR_TN_TR=0 PAPR_GEMM_MODE=h1 python3 scripts/probes/chain_bench.py 2>/dev/null; } > $O/cb.txt 2>&1; cat $O/cb.txt
run() { PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 20 --warmup 5 --no-amp-line --no-shipped-line --psnr-steps 0 --no-cpu-baseline "$@" 2> $O/l.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(j['ms_per_step'],3), j['config']['final_loss'])"; grep "^kernel  *8 " $O/l.txt | head -5; }
{ echo "=== default"; run; echo "=== no tr"; PAPR_TN_TR=0 run; echo "=== default"; run; echo "=== amp"; run --amp; echo "=== amp no tr"; PAPR_TN_TR=0 run --amp; } > $O/bench.txt 2>&1; cat $O/bench.txt
