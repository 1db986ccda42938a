"""Times of the fused-run launches alone: 4 x (512,000 x 256 x 256): training forward (all layers stored), data-gradient run,
inference (last layer stored); from the library's own HIP-event records.   PAPR_C4_FUSED=0: two-role slots only; PAPR_GEMM_MODE=h1: one-product mode; BENCH_M: rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papr_amd import ops, hip
M, n = int(os.environ.get('BENCH_M', '512000')), 4
d = torch.device("cuda:0")
spec = ops.MlpSpec("b", 256, dict(n_ff_layer=n, d_ff=256, d_ff_out=256, norm="none", ff_act="relu", ff_last_act="none"))
g = torch.Generator().manual_seed(0)
ws = [(torch.randn(256, 256, generator=g) * 0.1).to(d) for _ in range(n)]
bs = [(torch.randn(256, generator=g) * 0.05).to(d) for _ in range(n)]
x = torch.randn(M, 256, generator=g).to(d)
gy = torch.randn(M, 256, generator=g).to(d)
scratch = [torch.empty((M, 256), device=d) for _ in range(2)]
def once():
    outs = ops.mlp_forward(spec, ws, bs, x, M, keep=True)
    ops.mlp_backward(spec, ws, bs, x, M, outs, gy.clone(), scratch, True)
    ops.mlp_forward(spec, ws, bs, x, M, keep=False)
for _ in range(2): once()
torch.cuda.synchronize()
hip.profile_enable(True)
for _ in range(5): once()
torch.cuda.synchronize()
hip.profile_enable(False)
recs = hip.profile_collect()
fw = [r[4] for r in recs if r[0] == 9]
dg = [r[4] for r in recs if r[0] == 10]
wg = [r[4] for r in recs if r[0] == 8]
med = lambda v: sorted(v)[len(v) // 2] * 1e3 if v else float("nan")
print("M=%d " % M + "fused=%s mode=%s  4-layer run, us: training forward %.0f  data-gradient %.0f  inference %.0f   (weight-gradient batch %.0f)" %
      (os.environ.get("PAPR_C4_FUSED", "1"), os.environ.get("PAPR_GEMM_MODE", "h3"), med(fw[0::2]), med(dg), med(fw[1::2]), med(wg)))
