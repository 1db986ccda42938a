"""Cycle stamps of one persistent workgroup of the split-f16 GEMM (library built with -DPAPR_H3_TRACE)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papr_amd import ops, hip
M = 512000
d = torch.device("cuda:0")
spec = ops.MlpSpec("b", 256, dict(n_ff_layer=2, d_ff=256, d_ff_out=256, norm="none", ff_act="relu", ff_last_act="none"))
ws = [(torch.randn(256, 256) * 0.1).to(d) for _ in range(2)]
bs = [torch.zeros(256, device=d) for _ in range(2)]
x = torch.randn(M, 256, device=d)
for _ in range(3):
    ops.mlp_forward(spec, ws, bs, x, M, keep=True)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 256)()
hip.lib().papr_h3_trace_read(buf)
t = list(buf)
print("prologue (first loads, split, barrier):", t[1] - t[0], "(includes first compute)")
print(" g: compute  split+store  issue-load  epilogue  barrier")
for g in range(0, 40):
    b = 1 + g * 5
    prev = t[b - 1] if g else t[0]
    print("%2d: %7d %9d %9d %9d %8d" % (g, t[b] - prev, t[b + 1] - t[b], t[b + 2] - t[b + 1], t[b + 3] - t[b + 2], t[b + 4] - t[b + 3]))
