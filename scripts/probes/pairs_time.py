"""Wall time of papr_group_pairs on a step's shape (R = 25,600 rays, k = 20, P = 10,000 / 30,000)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papr_amd import ops
d = torch.device("cuda:0")
for P in (10000, 30000):
    R, k = 25600, 20
    g = torch.Generator().manual_seed(P)
    base = (torch.arange(R) * 3) % (P - 2 * k)
    flat = (base[:, None] + torch.stack([torch.randperm(2 * k, generator=g)[:k] for _ in range(R)])).reshape(-1).int().to(d)
    for _ in range(3): ops.group_pairs(flat, P, run=k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.group_pairs(flat, P, run=k)
    e1.record(); torch.cuda.synchronize()
    print("P=%d: %.1f us per call" % (P, e0.elapsed_time(e1) * 1e3 / 20))
