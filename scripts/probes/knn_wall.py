"""ray_knn wall time per call (events around 20 calls: ray packing, binning and the search itself), P from argv."""
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
for P in ([int(a) for a in sys.argv[1:]] or [10000, 30000]):
    torch.manual_seed(0)
    pd = ((torch.rand(P, 3) * 2 - 1) * 12).to(d)
    for _ in range(3):
        ops.ray_knn(pd, rayo, rd, 25600, 20, 1e-6)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.ray_knn(pd, rayo, rd, 25600, 20, 1e-6)
    e1.record(); torch.cuda.synchronize()
    print("P=%d  T=%s blocks=%s: %.1f us per call (wall, all kernels)" % (P, os.environ.get("PAPR_KNN_T", "auto"), os.environ.get("PAPR_KNN_BLOCKS", "1"), e0.elapsed_time(e1) * 1e3 / 20))
