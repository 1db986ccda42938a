"""Weight-gradient launch (gemm_tn_h3, kernel id 8) of a 5-layer 256-wide MLP at the chair shape in the parity mode (fp32 rows) and in the
one-product mode (f16 rows): time, algorithmic bytes, GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papr_amd import ops, hip

M = int(sys.argv[1]) if len(sys.argv) > 1 else 512000
d = torch.device("cuda:0")
n = 5
ecfg = dict(n_ff_layer=n, d_ff=256, d_ff_out=256, norm="none", ff_act="relu", ff_last_act="none")
g = torch.Generator(device="cpu").manual_seed(0)
ws = [((torch.rand(256, 256, generator=g) * 2 - 1) * 0.108).to(d) for _ in range(n)]
bs = [((torch.rand(256, generator=g) * 2 - 1) * 0.05).to(d) for _ in range(n)]
x = torch.randn(M, 256, device=d)
gy = torch.randn(M, 256, device=d)
scratch = [torch.empty((M, 256), device=d) for _ in range(2)]
for one in (False, True):
    spec = ops.MlpSpec("b", 256, ecfg)
    spec.one_product = one
    for rep in range(2):
        outs = ops.mlp_forward(spec, ws, bs, x, M, keep=True)
        ops.mlp_backward(spec, ws, bs, x, M, outs, gy.clone(), scratch, True)
    torch.cuda.synchronize()
    hip.profile_enable(True)
    for rep in range(5):
        outs = ops.mlp_forward(spec, ws, bs, x, M, keep=True)
        ops.mlp_backward(spec, ws, bs, x, M, outs, gy.clone(), scratch, True)
    hip.profile_enable(False)
    recs = hip.profile_collect()
    for kid in (8, 9, 10):
        r = [t for t in recs if t[0] == kid]
        if not r:
            continue
        ms = sum(t[4] for t in r) / len(r)
        by = sum(t[5] for t in r) / len(r)
        print("one_product=%d kernel %2d: n=%d avg %.1f us, %.2f GB algorithmic -> %.2f TB/s" % (one, kid, len(r), ms * 1e3, by / 1e9, by / ms / 1e9))
