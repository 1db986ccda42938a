"""Blocks visited / insertions / offers per ray of the spatial ray_knn (library built with -DKNN_DEBUG_VISITS: PAPR_HIP_LIB=scripts/probes/bin/libpapr_knndbg.so)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papr_amd import ops, load_config
from papr_amd.data import SyntheticRayData
cfg = load_config("nerfsyn/chair.yml")
d = torch.device("cuda:0")
data = SyntheticRayData(cfg["dataset"], n_views=4, seed=1, device=d)
_, rayd, rayo, _ = data.patch()
rd = rayd.reshape(-1, 3).contiguous()
T = int(os.environ.get("PAPR_KNN_T", "5"))
for P in (10000, 30000):
    torch.manual_seed(0)
    pd = ((torch.rand(P, 3) * 2 - 1) * 12).to(d)
    idx, dist = ops.ray_knn(pd, rayo, rd, 25600, 20, 1e-6, want_dist=True)
    v = dist.view(-1, 20)[:, :3].cpu()
    cold = torch.arange(v.shape[0]) % T == 0
    for name, m in (("cold", cold), ("warm", ~cold)):
        print("P=%d %s rays: blocks visited mean %.1f max %.0f of %d | insertions mean %.1f | offers mean %.1f" %
              (P, name, v[m, 0].mean(), v[m, 0].max(), (P + 63) // 64, v[m, 1].mean(), v[m, 2].mean()))
