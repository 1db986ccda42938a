"""The two forms of ray_knn on the same inputs in ONE process: the library is loaded twice (a copy under another name), the copy's first call is
made with PAPR_KNN_BLOCKS=0.  Many patches of real camera rays against clouds like a trained one; prints rays whose index lists differ."""
import ctypes as C, os, shutil, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from papr_amd import hip, ops, load_config
from papr_amd.data import SyntheticRayData
cfg = load_config("nerfsyn/chair.yml")
d = torch.device("cuda:0")
data = SyntheticRayData(cfg["dataset"], n_views=20, seed=1, device=d)
libA = hip.lib()
tmp = os.path.join(tempfile.mkdtemp(), "libpapr_every_point.so"); shutil.copy(hip.LIB_PATH, tmp)
libB = C.CDLL(tmp)
for L in (libB,):
    L.papr_ray_knn_workspace_bytes.restype = C.c_size_t
    L.papr_ray_knn_workspace_bytes.argtypes = libA.papr_ray_knn_workspace_bytes.argtypes
    L.papr_ray_knn.argtypes = libA.papr_ray_knn.argtypes
def knn(L, pts, ro, rd, rpi, k):
    R = rd.shape[0]
    idx = torch.empty((R, k), device=d, dtype=torch.int32); dist = torch.empty((R, k), device=d)
    ws = torch.empty(L.papr_ray_knn_workspace_bytes(R, pts.shape[0]) // 4, device=d)
    rc = L.papr_ray_knn(hip.ptr(pts), pts.shape[0], hip.ptr(ro), hip.ptr(rd), R, rpi, k, 1e-6, hip.ptr(idx), hip.ptr(dist), hip.ptr(ws), hip.stream_ptr())
    assert rc == 0
    return idx, dist
g = torch.Generator().manual_seed(3)
def cloud(P, kind):
    if kind == "cube": return ((torch.rand(P, 3, generator=g) * 2 - 1) * 1.2 * 10).to(d)
    v = torch.randn(P, 3, generator=g); s = 6 * v / v.norm(dim=1, keepdim=True) * (0.6 + 0.4 * torch.rand(P, 1, generator=g) ** 3)
    return (s + 0.01 * torch.randn(P, 3, generator=g)).to(d).contiguous()
os.environ["PAPR_KNN_BLOCKS"] = "0"
first = True
tot = bad = 0
for kind in ("cube", "shell"):
    for P in (2100, 2908, 4908, 10000, 11405):
        pts = cloud(P, kind)
        for rep in range(6):
            _, rayd, rayo, _ = data.patch()
            rd = rayd.reshape(-1, 3).contiguous()
            ib, db = knn(libB, pts, rayo, rd, 25600, 20)          # every point (its first call reads the environment)
            if first: os.environ.pop("PAPR_KNN_BLOCKS"); first = False
            ia, da = knn(libA, pts, rayo, rd, 25600, 20)
            ia2, _ = knn(libA, pts, rayo, rd, 25600, 20)
            torch.cuda.synchronize()
            diff = (ia != ib).any(-1); rerun = (ia != ia2).any(-1)
            n = int(diff.sum()); tot += ia.shape[0]
            # a difference is fine only at an exact tie of the k-th distance
            notie = int((diff & (da[:, -1] != db[:, -1])).sum())
            wrong = int((diff & (da[:, -1] > db[:, -1])).sum())
            bad += notie
            print("%-5s P=%5d patch %d: %d rays differ (%d with another k-th distance, %d with a LARGER one), %d differ between two runs of the spatial form" % (kind, P, rep, n, notie, wrong, int(rerun.sum())))
print("rays compared:", tot, "not ties:", bad)
