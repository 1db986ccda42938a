"""ray_knn (spatial form) against the oracle on awkward clouds: tight clusters, sheets, far outliers, rays from inside the cloud, after-prune
sizes.  Prints one line per case; exits non-zero on a set that differs from the oracle's other than at an exact tie of the k-th distance."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from papr_amd import ops
import oracle.papr_oracle as O

d = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
def cloud(kind, P):
    if kind == "uniform": return (torch.rand(P, 3, generator=g) * 2 - 1) * 12
    if kind == "clusters":
        c = (torch.rand(12, 3, generator=g) * 2 - 1) * 10
        return c[torch.randint(0, 12, (P,), generator=g)] + torch.randn(P, 3, generator=g) * 0.05
    if kind == "sheet":
        p = (torch.rand(P, 3, generator=g) * 2 - 1) * 8; p[:, 2] = 0.3 * torch.sin(p[:, 0]) + 1e-3 * torch.randn(P, generator=g); return p
    if kind == "outliers":
        p = torch.randn(P, 3, generator=g) * 2; p[: P // 200] *= 500.0; return p
    if kind == "surface":       # a sphere shell, like a trained cloud
        v = torch.randn(P, 3, generator=g); return 5 * v / v.norm(dim=1, keepdim=True) + 0.02 * torch.randn(P, 3, generator=g)
    if kind == "line":          # everything near one line
        t = torch.rand(P, 1, generator=g) * 20 - 10; return t * torch.tensor([[0.6, 0.64, 0.48]]) + 1e-3 * torch.randn(P, 3, generator=g)
bad = 0
for kind in ("uniform", "clusters", "sheet", "outliers", "surface", "line"):
    for P in (2048, 2900, 5000, 11111, 30000):
        for origin in ("outside", "inside"):
            pts = cloud(kind, P)
            n = 96
            o = torch.tensor([[0.0, -14.0, 3.0]]) if origin == "outside" else pts[:1].clone() + 0.01
            dirs = torch.randn(n, 3, generator=g)
            if origin == "outside": dirs = (pts[torch.randint(0, P, (n,), generator=g)] - o) + 0.3 * torch.randn(n, 3, generator=g)
            k = 20
            idx, dist = ops.ray_knn(pts.to(d), o.to(d), dirs.contiguous().to(d), n, k, 1e-6, want_dist=True)
            feat = O.ray_point_distance(pts, o, dirs.view(1, n, 1, 3), 1e-6).reshape(n, P)
            ref = feat.topk(k, largest=False)
            got = np.sort(idx.cpu().numpy(), -1); want = np.sort(ref.indices.numpy(), -1)
            mism = (got != want).any(-1)
            d_got = torch.gather(feat, 1, idx.cpu().long())
            tie_ok = True
            if mism.any():
                tie_ok = bool(torch.all(d_got.max(-1).values[torch.from_numpy(mism)] <= ref.values.max(-1).values[torch.from_numpy(mism)] * (1 + 2e-6) + 1e-12))
            asc = bool(torch.all(d_got[:, 1:] >= d_got[:, :-1] - 1e-6 * (1 + d_got[:, 1:].abs())))
            ok = tie_ok and asc
            bad += 0 if ok else 1
            print("%-9s P=%6d %-7s: %3d of %d rays differ from the oracle's sets%s%s" % (kind, P, origin, int(mism.sum()), n, "" if tie_ok else "  NOT TIES", "" if asc else "  NOT ASCENDING"))
sys.exit(1 if bad else 0)
