#!/bin/bash
# round 6: the parity form of the same (papr_f16_rows with a lo plane)
O=gpurun_out/r6t2; mkdir -p $O
{ timeout 900 python3 -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "attention_tail" 2>&1 | tail -5
  timeout 1500 python3 -m pytest tests/test_hip_chain_variants.py -q -m gpu -x 2>&1 | tail -5
  timeout 1500 python3 -m pytest tests/test_hip_model.py -q -m gpu -x 2>&1 | tail -5
} > $O/tests.txt 2>&1
for v in 1 0 1 0; do
  echo "PAPR_TAIL_F16=$v" >> $O/bench.txt
  PAPR_TAIL_F16=$v timeout 600 python3 bench.py --no-cpu-baseline --no-amp-line --no-shipped-line --steps 30 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $O/bench.txt 2>&1
done
cat $O/tests.txt $O/bench.txt
