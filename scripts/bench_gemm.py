#!/usr/bin/env python3
"""Micro-benchmark of the MLP GEMM kernels at the chair shape (M = 512,000 rows, 256-wide layers).
Prints per-kernel-id average launch time and TFLOP/s from the library's HIP-event records."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from papr_amd import ops, hip

M = int(sys.argv[1]) if len(sys.argv) > 1 else 512000
d = torch.device("cuda:0")
ecfg = dict(n_ff_layer=4, d_ff=256, d_ff_out=256, norm="none", ff_act="relu", ff_last_act="none")
spec = ops.MlpSpec("b", 256, ecfg)
g = torch.Generator(device="cpu").manual_seed(0)
ws = [((torch.rand(256, 256, generator=g) * 2 - 1) * 0.108).to(d) for _ in range(4)]
bs = [((torch.rand(256, generator=g) * 2 - 1) * 0.05).to(d) for _ in range(4)]
x = torch.randn(M, 256, device=d)
gy = torch.randn(M, 256, device=d)
scratch = [torch.empty((M, 256), device=d) for _ in range(2)]
names = {0: "gemm_nt<128x256>", 4: "gemm_tn"}
for rep in range(3):
    outs = ops.mlp_forward(spec, ws, bs, x, M, keep=True)
    ops.mlp_backward(spec, ws, bs, x, M, outs, gy.clone(), scratch, True)
torch.cuda.synchronize()
hip.profile_enable(True)
for rep in range(5):
    outs = ops.mlp_forward(spec, ws, bs, x, M, keep=True)
hip.profile_enable(False)
fwd = hip.profile_collect()
hip.profile_enable(True)
for rep in range(5):
    ops.mlp_backward(spec, ws, bs, x, M, outs, gy.clone(), scratch, True)
hip.profile_enable(False)
bwd = hip.profile_collect()
def rep(tag, recs, kid):
    r = [x for x in recs if x[0] == kid]
    if not r: return
    ms = sum(x[4] for x in r) / len(r)
    fl = sum(2.0 * x[1] * x[2] * x[3] for x in r) / len(r)
    print("%-28s n=%3d avg %.1f us  %.1f TFLOP/s" % (tag, len(r), ms * 1e3, fl / ms / 1e9))
rep("fwd  gemm_nt", fwd, 0)
rep("dgrad gemm_nt", bwd, 0)
rep("wgrad gemm_tn", bwd, 4)
rep("fwd  gemm_nt_h3", fwd, 6)
rep("dgrad gemm_nt_h3", bwd, 7)
r8 = [x for x in bwd if x[0] == 8]
if r8:      # one record per batch of weight-gradients: N = jobs in the launch, bytes / flops summed over them
    ms = sum(x[4] for x in r8); jobs = sum(x[2] for x in r8)
    print("%-28s n=%3d launches of %d jobs, avg %.1f us per job  %.0f GB/s  %.1f TFLOP/s" % ("wgrad gemm_tn_h3", len(r8), r8[0][2], ms * 1e3 / jobs,
          sum(x[5] for x in r8) / ms / 1e6, sum(x[6] for x in r8) / ms / 1e9))
def rep_chain(tag, recs, kid):
    r = [x for x in recs if x[0] == kid]
    if not r: return
    ms = sum(x[4] for x in r) / len(r)
    print("%-28s n=%3d avg %.1f us per launch of %d layers (%.0f GB/s algorithmic, %.0f TFLOP/s of f16 MFMA = %.0f fp32-equivalent)"
          % (tag, len(r), ms * 1e3, r[0][2], r[0][5] / ms / 1e6, 3 * r[0][6] / ms / 1e9, r[0][6] / ms / 1e9))
rep_chain("fwd  mlp_chain", fwd, 9)
rep_chain("dgrad mlp_chain", bwd, 10)
hip.profile_enable(True)
for rep_ in range(5):
    ops.mlp_forward(spec, ws, bs, x, M, keep=False)
hip.profile_enable(False)
rep_chain("inference mlp_chain", hip.profile_collect(), 9)
