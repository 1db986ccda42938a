"""What a one-product data-gradient run makes of top gradient rows that hold inf / nan (the GradScaler's overflow signal must survive), in a fresh process
(the mode is read when the library loads):   python3 tests/inf_rows_worker.py      -- tests/test_hip_chain_variants.py reads the printed lines"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PAPR_GEMM_MODE"] = "h1"
import torch
from papr_amd import ops
M, n = 4096, 4
d = torch.device("cuda:0")
spec = ops.MlpSpec("b", 256, dict(n_ff_layer=n, d_ff=256, d_ff_out=256, norm="none", ff_act="relu", ff_last_act="none"))
g = torch.Generator().manual_seed(0)
ws = [(torch.randn(256, 256, generator=g) * 0.1).to(d) for _ in range(n)]
bs = [(torch.randn(256, generator=g) * 0.05).to(d) for _ in range(n)]
x = torch.randn(M, 256, generator=g).to(d)
scratch = [torch.empty((M, 256), device=d) for _ in range(2)]
for label, fill in (("all inf", float("inf")), ("one inf per row", None), ("all nan", float("nan")), ("huge finite 1e38", 1e38)):
    gy = torch.randn(M, 256, generator=g).to(d)
    if fill is None: gy[:, 7] = float("inf")
    else: gy[:] = fill
    outs = ops.mlp_forward(spec, ws, bs, x, M, keep=True)
    d_ws, d_bs, d_x = ops.mlp_backward(spec, ws, bs, x, M, outs, gy.clone(), scratch, True)
    torch.cuda.synchronize()
    print(label, "| d_x finite elements: %d of %d" % (int(torch.isfinite(d_x).sum()), d_x.numel()), "| d_w finite:", [int(torch.isfinite(t).sum()) for t in d_ws])
