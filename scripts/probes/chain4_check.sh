#!/bin/bash
# the fused-run kernel on the GPU box: its forms bit for bit against each other, then the isolated 4-layer timings (parity mode and h1)
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_hip_chain_variants.py -x -q 2>&1 | tail -15 > gpurun_out/c4_variants.txt
{ python3 scripts/probes/chain_bench.py; PAPR_C4_FUSED=0 python3 scripts/probes/chain_bench.py; PAPR_GEMM_MODE=h1 python3 scripts/probes/chain_bench.py; } > gpurun_out/c4_bench.txt 2>&1
cat gpurun_out/c4_variants.txt gpurun_out/c4_bench.txt
