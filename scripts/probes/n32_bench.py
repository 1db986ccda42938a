"""Times of the weight-gradient launches of a 256 -> 256 -> 256 -> 32 MLP (the last one: gemm_tn_tr_n32_kernel), from the library's HIP-event records."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papr_amd import ops, hip
M, n = int(os.environ.get('BENCH_M', '512000')), 3
d = torch.device("cuda:0")
spec = ops.MlpSpec("b", 256, dict(n_ff_layer=n, d_ff=256, d_ff_out=32, norm="none", ff_act="relu", ff_last_act="none"))
g = torch.Generator().manual_seed(0)
ws = [(torch.randn(32 if i == n - 1 else 256, 256, generator=g) * 0.1).to(d) for i in range(n)]
bs = [(torch.randn(32 if i == n - 1 else 256, generator=g) * 0.05).to(d) for i in range(n)]
x = torch.randn(M, 256, generator=g).to(d)
gy = torch.randn(M, spec.ld_out[-1], generator=g).to(d)
scratch = [torch.empty((M, 256), device=d) for _ in range(2)]
def once():
    outs = ops.mlp_forward(spec, ws, bs, x, M, keep=True)
    ops.mlp_backward(spec, ws, bs, x, M, outs, gy.clone(), scratch, True)
for _ in range(2): once()
torch.cuda.synchronize()
hip.profile_enable(True)
for _ in range(5): once()
torch.cuda.synchronize()
hip.profile_enable(False)
recs = [r for r in hip.profile_collect() if r[0] == 8]
per = len(recs) // 5
for i in range(per):
    v = sorted(r[4] for r in recs[i::per])
    print("weight-gradient launch %d: jobs %d  %.0f us" % (i, recs[i][2], v[len(v) // 2] * 1e3))
