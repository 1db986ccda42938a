#!/bin/bash
# round 6, GPU call 4: the one-product mode with one scale per row and run (ONE v2) -- forms, tolerances, AMP goldens, times
O=gpurun_out/r6j; mkdir -p $O
timeout 900 python3 -m pytest tests/test_hip_chain_variants.py -q -m gpu -k "one_product" -x -s > $O/variants.txt 2>&1; echo "variants rc $?" >> $O/rc.txt
timeout 900 python3 -m pytest tests/test_hip_h1.py -q -m gpu -x -s > $O/h1.txt 2>&1; echo "h1 rc $?" >> $O/rc.txt
timeout 900 python3 -m pytest tests/test_hip_amp_golden.py -q -m gpu -s > $O/amp_golden.txt 2>&1; echo "amp golden rc $?" >> $O/rc.txt
timeout 900 python3 -m pytest tests/test_hip_model.py -q -m gpu -k "amp" > $O/model_amp.txt 2>&1; echo "model amp rc $?" >> $O/rc.txt
run() { PAPR_BENCH_LAUNCHES=1 python3 bench.py --steps 20 --warmup 5 --no-amp-line --no-shipped-line --psnr-steps 0 --no-cpu-baseline "$@" 2> $O/l.txt | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', round(j['ms_per_step'],3), 'final_loss', j['config'].get('final_loss'))"; grep "^kernel  *\(9\|10\|8\) " $O/l.txt | head -8; }
{ echo "=== amp"; run --amp; echo "=== amp prev lib"; PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_prev.so run --amp; echo "=== amp again"; run --amp; echo "=== amp prev lib again"; PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_prev.so run --amp; echo "=== h1"; run --gemm-mode h1; echo "=== default"; run; echo "=== fp32 rows"; run --h3-rows f32; } > $O/bench_amp.txt 2>&1
{ echo "h1:"; PAPR_GEMM_MODE=h1 python3 scripts/probes/chain_bench.py 2>/dev/null; echo "h1 single slots:"; PAPR_C4_PAIRS=0 PAPR_GEMM_MODE=h1 python3 scripts/probes/chain_bench.py 2>/dev/null; echo "h1 two-role:"; PAPR_C4_FUSED=0 PAPR_GEMM_MODE=h1 python3 scripts/probes/chain_bench.py 2>/dev/null; } > $O/chain_bench.txt 2>&1
timeout 1700 python3 -m pytest tests -q -m gpu -x > $O/suite.txt 2>&1; echo "suite rc $?" >> $O/rc.txt
tail -3 $O/suite.txt; cat $O/rc.txt; cat $O/bench_amp.txt | grep "ms_per\|==="; cat $O/chain_bench.txt
