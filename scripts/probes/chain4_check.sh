#!/bin/bash
# chain4.hip on the GPU box: bit-for-bit against the other fused-run kernels, then the isolated 4-layer timings of both.
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_hip_chain_variants.py -x -q 2>&1 | tail -15 > gpurun_out/c4_variants.txt
for v in 3 4; do PAPR_CHAIN=$v timeout 300 python3 scripts/probes/chain_bench.py; done > gpurun_out/c4_bench.txt 2>&1
PAPR_CHAIN=4 PAPR_GEMM_MODE=h1 timeout 300 python3 scripts/probes/chain_bench.py >> gpurun_out/c4_bench.txt 2>&1
PAPR_CHAIN=3 PAPR_GEMM_MODE=h1 timeout 300 python3 scripts/probes/chain_bench.py >> gpurun_out/c4_bench.txt 2>&1
cat gpurun_out/c4_variants.txt gpurun_out/c4_bench.txt
