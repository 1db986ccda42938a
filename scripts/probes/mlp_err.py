#!/usr/bin/env python3
"""Per-layer gradient error of the MLP chain against torch fp64 (diagnostic).  usage: mlp_err.py M d_in width d_out n"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papr_amd import ops
M, d_in, width, d_out, n = [int(a) for a in sys.argv[1:6]]
gen = torch.Generator().manual_seed(M)
ecfg = dict(n_ff_layer=n, d_ff=width, d_ff_out=d_out, norm="none", ff_act="relu", ff_last_act="none", skip_layers=[])
spec = ops.MlpSpec("t", d_in, ecfg)
ws, bs = [], []
for i in range(n):
    fi = d_in if i == 0 else width
    fo = d_out if i == n - 1 else width
    ws.append(((torch.rand(fo, fi, generator=gen) * 2 - 1) * (6.0 / (fi + fo)) ** 0.5))
    bs.append(((torch.rand(fo, generator=gen) * 2 - 1) * 0.1))
x = torch.randn(M, d_in, generator=gen)
gy = torch.randn(M, d_out, generator=gen)
def ref(dt):
    W = [w.to(dt).requires_grad_(True) for w in ws]; B = [b.to(dt).requires_grad_(True) for b in bs]
    X = x.to(dt).requires_grad_(True); h = X
    for i in range(n):
        h = torch.nn.functional.linear(h, W[i], B[i])
        if i < n - 1: h = torch.relu(h)
    (h * gy.to(dt)).sum().backward()
    return h.detach(), [w.grad for w in W], [b.grad for b in B], X.grad
y64, dw64, db64, dx64 = ref(torch.float64)
y32, dw32, db32, dx32 = ref(torch.float32)
d = torch.device("cuda:0")
wd = [w.to(d) for w in ws]; bd = [b.to(d) for b in bs]
ew, eb = ops.prepare_mlp_weights(spec, wd, bd)
xp = torch.zeros(M, spec.ld_in); xp[:, :d_in] = x
xd = xp.to(d)
outs = ops.mlp_forward(spec, ew, eb, xd, M, keep=True)
gp = torch.zeros(M, spec.ld_out[-1]); gp[:, :d_out] = gy
wmax = max(width, spec.ld_in, spec.ld_out[-1])
scratch = [torch.empty((M, wmax), device=d) for _ in range(2)]
d_ws, d_bs, d_x = ops.mlp_backward(spec, ew, eb, xd, M, outs, gp.to(d), scratch, True)
torch.cuda.synchronize()
def rel(a, b): return ((a.double() - b).abs().max() / b.abs().max()).item()
print("mode", os.environ.get("PAPR_GEMM_MODE", "h3"), "M", M)
print("  y    hip %.2e   torch32 %.2e" % (rel(outs[-1].cpu()[:, :d_out], y64), rel(y32, y64)))
for i in range(n):
    fi = d_in if i == 0 else width
    fo = d_out if i == n - 1 else width
    print("  dW%d  hip %.2e   torch32 %.2e    db hip %.2e torch32 %.2e" % (i, rel(d_ws[i].cpu()[:fo, :fi], dw64[i]), rel(dw32[i], dw64[i]),
                                                                        rel(d_bs[i].cpu()[:fo], db64[i]), rel(db32[i], db64[i])))
print("  dx   hip %.2e   torch32 %.2e" % (rel(d_x.cpu()[:, :d_in], dx64), rel(dx32, dx64)))
