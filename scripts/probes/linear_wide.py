"""Does ops.linear_rows (one layer through papr_mlp_fwd) take wide / deep GEMMs?  (the 2x2 transposed convolution as four taps)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papr_amd import ops
d = torch.device("cuda:0")
for M, K, N in [(1600, 512, 1024), (1600, 1024, 512), (6400, 256, 512), (6400, 512, 256)]:
    x = torch.randn(M, K, device=d); w = torch.randn(N, K, device=d) / K ** 0.5
    try:
        y = ops.linear_rows(x, w)
        torch.cuda.synchronize()
        ref = (x.double() @ w.double().t()).float()
        t0 = time.perf_counter()
        for _ in range(20): ops.linear_rows(x, w)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for _ in range(20): x @ w.t()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(M, K, N, "max err %.2e" % (y - ref).abs().max().item(), "own %.1f us  torch %.1f us" % ((t1 - t0) / 20 * 1e6, (t2 - t1) / 20 * 1e6))
    except Exception as e:
        print(M, K, N, "FAILED", str(e)[:200])
