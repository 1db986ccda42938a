"""Runs only the fused forward chain (inference mode) a few times: target for rocprofv3 --pmc passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papr_amd import ops
M = 512000
d = torch.device("cuda:0")
n = 4
spec = ops.MlpSpec("b", 256, dict(n_ff_layer=n, d_ff=256, d_ff_out=256, norm="none", ff_act="relu", ff_last_act="none"))
ws = [(torch.randn(256, 256) * 0.1).to(d) for _ in range(n)]
bs = [torch.zeros(256, device=d) for _ in range(n)]
x = torch.randn(M, 256, device=d)
keep = len(sys.argv) > 1 and sys.argv[1] == "keep"
for _ in range(4):
    ops.mlp_forward(spec, ws, bs, x, M, keep=keep)
torch.cuda.synchronize()
