"""one-product mode against fp64 torch on a leaky / relu MLP: forward error and gradient errors (is the LeakyReLU path of round 6's one-product arithmetic off?)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papr_amd import ops
act = sys.argv[1] if len(sys.argv) > 1 else "leakyrelu"
M, n, d_in, width, d_out = 20000, 5, 117, 256, 256
gen = torch.Generator().manual_seed(3)
spec = ops.MlpSpec("t", d_in, dict(n_ff_layer=n, d_ff=width, d_ff_out=d_out, norm="none", ff_act=act, ff_last_act="none", skip_layers=[]))
spec.one_product = os.environ.get("ONE", "1") == "1"
d = torch.device("cuda:0")
ws, bs = [], []
for i in range(n):
    fi = d_in if i == 0 else width
    w = torch.zeros(width if i < n - 1 else d_out, spec.layers[i]["n_in"])
    w[:, :fi] = (torch.rand(w.shape[0], fi, generator=gen) * 2 - 1) * (6.0 / (fi + w.shape[0])) ** 0.5
    ws.append(w); bs.append((torch.rand(w.shape[0], generator=gen) * 2 - 1) * 0.1)
x = torch.zeros(M, spec.ld_in); x[:, :d_in] = torch.randn(M, d_in, generator=gen)
gy = torch.randn(M, d_out, generator=gen)
# fp64 reference
W = [w.double().requires_grad_(True) for w in ws]; B = [b.double().requires_grad_(True) for b in bs]
h = x.double().requires_grad_(True); hh = h
for i in range(n):
    hh = torch.nn.functional.linear(hh, W[i], B[i])
    if i < n - 1: hh = torch.relu(hh) if act == "relu" else torch.nn.functional.leaky_relu(hh, 0.2)
(hh * gy.double()).sum().backward()
wd = [w.to(d) for w in ws]; bd = [b.to(d) for b in bs]; xd = x.to(d)
outs = ops.mlp_forward(spec, wd, bd, xd, M, keep=True)
scratch = [torch.empty((M, 256), device=d) for _ in range(2)]
d_ws, d_bs, d_x = ops.mlp_backward(spec, wd, bd, xd, M, outs, gy.to(d).clone(), scratch, True)
def r(a, b): return float((a.double().cpu() - b).pow(2).mean().sqrt() / b.abs().max())
print(act, "ONE" if spec.one_product else "parity", "y %.2e  d_x %.2e" % (r(outs[-1], hh.detach()), r(d_x, h.grad)), " dW", " ".join("%.2e" % r(a, b.grad) for a, b in zip(d_ws, W)), " db", " ".join("%.2e" % r(a, b.grad) for a, b in zip(d_bs, B)))
