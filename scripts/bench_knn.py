#!/usr/bin/env python3
"""ray_knn timing: lattice-ordered vs shuffled point order, P = 10k / 30k, R = 25,600 rays."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from papr_amd import ops, hip
from papr_amd.data import SyntheticRayData
from papr_amd import load_config
cfg = load_config("nerfsyn/chair.yml")
d = torch.device("cuda:0")
data = SyntheticRayData(cfg["dataset"], n_views=4, seed=1, device=d)
_, rayd, rayo, _ = data.patch()
rd = rayd.reshape(-1, 3).contiguous()
def lattice(P):
    n = int(P ** (1 / 3.0))
    ax = torch.linspace(-12, 12, n)
    g = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    rest = (torch.rand(P - g.shape[0], 3) * 2 - 1) * 12
    return torch.cat([g, rest])
for P in ([int(a) for a in sys.argv[1:]] or [10000, 30000]):
    pts = lattice(P)
    for tag, p in (("lattice order", pts), ("shuffled", pts[torch.randperm(P)])):
        pd = p.to(d).contiguous()
        for _ in range(3):
            ops.ray_knn(pd, rayo, rd, 25600, 20, 1e-6)
        torch.cuda.synchronize()
        hip.profile_enable(True)
        for _ in range(10):
            ops.ray_knn(pd, rayo, rd, 25600, 20, 1e-6)
        hip.profile_enable(False)
        r = [x[4] for x in hip.profile_collect() if x[0] == 5]
        print("P=%d %-14s ray_knn avg %.1f us" % (P, tag, 1e3 * sum(r) / len(r)))
