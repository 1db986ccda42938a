"""Generate the golden fixtures by running the REFERENCE (zvict/papr, pure Python) on CPU.

Run only in the build container:   python tests/golden/make_golden.py
It needs /root/reference (read-only) and writes tests/golden/*.npz / *.json.  Nothing from
the reference is copied: the fixtures hold seeded inputs and the reference's numeric outputs.

Process-local shims: stub modules for lpips/torchvision/imageio (absent,
never touched by the render path) and an lr_scheduler wrapper that swallows the `verbose=`
keyword removed from recent torch.
"""
import copy
import json
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, REF)
for name in ["lpips", "torchvision", "torchvision.models", "imageio"]:
    sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].models = sys.modules["torchvision.models"]
from PIL import Image as _PILImage  # noqa: E402
sys.modules["imageio"].imread = lambda path: np.array(_PILImage.open(path))      # (the reference only calls imageio.imread)
import torch.optim.lr_scheduler as ls  # noqa: E402

for n in ["LinearLR", "CosineAnnealingLR", "ExponentialLR", "StepLR", "SequentialLR"]:
    c = getattr(ls, n)
    o = c.__init__
    c.__init__ = (lambda o: lambda self, *a, verbose=None, **k: o(self, *a, **k))(o)

# Shim 4 (round 5, for G17 = the reference under `use_amp: true`): the reference wraps its attention block, U-Net and mapping MLP in
# `autocast(device_type='cuda', ...)` (models/attn.py:248, models/unet.py:212, models/mlp.py:76) and builds `torch.cuda.amp.GradScaler`
# (models/model.py:26); without a CUDA device both switch themselves off.  Redirected to torch's CPU autocast / CPU GradScaler, process-locally,
# BEFORE the reference's modules bind `from torch import autocast`.  With `use_amp: false` (every other fixture) both stay disabled no-ops.
# What CPU autocast casts differently from the CUDA list: nothing that these three regions execute -- they run linear / addmm / matmul /
# conv2d / conv_transpose2d (fp16 on both lists), cat / stack (promote on both) and relu, max_pool2d, sin, cos, mean, std, add, mul, div
# (on neither list: input dtype on both).  The CUDA-only fp32 entries (pow, sum, softmax, norm, layer_norm, exp, log ...) are not called
# inside the regions (the reference's LayerNorm is hand-written from mean / std; softmax and the compositing run outside autocast).
# The fp16 kernels themselves differ (oneDNN on the host, rocBLAS / MIOpen on a GPU): both accumulate in fp32 and round the result to
# fp16 once, so the fixtures pin the reference's AMP arithmetic up to summation order.
_TorchAutocast = torch.autocast


class _CpuAutocast(_TorchAutocast):
    def __init__(self, device_type, dtype=None, enabled=True, cache_enabled=None):
        super().__init__("cpu" if device_type == "cuda" else device_type, dtype=dtype, enabled=enabled, cache_enabled=cache_enabled)


torch.autocast = _CpuAutocast
torch.cuda.amp.GradScaler = lambda enabled=True, **kw: torch.amp.GradScaler("cpu", enabled=enabled, **kw)

from utils import DictAsMember, update_dict, setup_seed  # noqa: E402  (reference utils.py)
from models import get_model, get_loss  # noqa: E402
from models.utils import posenc as ref_posenc  # noqa: E402
from models.attn import LayerNorm as RefLayerNorm  # noqa: E402
import train as ref_train  # noqa: E402

from formula import formula_fill, synth_rays, uniform_points, write_t2_fixture, write_blender_fixture  # noqa: E402

torch.set_num_threads(8)


def load_cfg(scene, **over):
    cfg = yaml.safe_load(open(REF + "/configs/default.yml"))
    update_dict(cfg, yaml.safe_load(open(REF + "/configs/" + scene)))
    cfg["use_amp"] = False
    cfg["training"]["losses"] = {"mse": 1.0, "lpips": 0.0, "lpips_alex": 0.0}
    update_dict(cfg, over)
    return cfg


def build(cfg):
    setup_seed(1)
    model = get_model(DictAsMember(copy.deepcopy(cfg)), "cpu")
    formula_fill(model.state_dict())
    return model


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), t.norm().item(), t.abs().max().item()])


def npf(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("wrote", name, "%.1f KB" % (os.path.getsize(path) / 1024))


# ----------------------------------------------------------------------------------- G1/G2
def g1_g2():
    g = torch.Generator().manual_seed(11)
    x = torch.cat([torch.randn(40, 3, generator=g) * 5.0,
                   torch.tensor([[50.0, -49.5, 0.0], [1e-3, -1e-4, 12.0], [33.3, 41.7, -38.9]])])
    out = {"x": npf(x)}
    for L in (4, 6):
        out["pe_L%d" % L] = npf(ref_posenc(x, L, 2.0, False, 1.0))
        out["pe_L%d_noself" % L] = npf(ref_posenc(x, L, 2.0, True, 1.0))
    for width in (39, 117, 256):
        ln = RefLayerNorm(width, 1e-6)
        xx = torch.randn(16, width, generator=g) * 3.0 + 0.5
        with torch.no_grad():
            ln.a_2.copy_(torch.rand(width, generator=g) + 0.5)
            ln.b_2.copy_(torch.rand(width, generator=g) - 0.5)
        out["ln%d_x" % width] = npf(xx)
        out["ln%d_a" % width] = npf(ln.a_2)
        out["ln%d_b" % width] = npf(ln.b_2)
        out["ln%d_y" % width] = npf(ln(xx))
    save("g12_posenc_layernorm.npz", **out)


# ----------------------------------------------------------------------------------- G3/G4
def g3_g4(model):
    out = {}
    # (a) reference cube init, P=1000, 16x16 rays
    pts = model.points.detach()
    ro, rd, c2w = synth_rays(1, 16, 16, seed=0)
    with torch.no_grad():
        idx = model._calculate_global_distances(ro, rd, pts)
        _, _, proj, D = model._calculate_distances(ro, rd, pts[idx], c2w)
    out.update(a_points=npf(pts), a_idx=np.sort(npf(idx), axis=-1).astype(np.int32),
               a_proj=npf(proj), a_D=npf(D), a_idx_raw=npf(idx).astype(np.int32))
    # (b) P=10000 uniform cloud, 32x32 rays (inputs regenerated from seeds in the tests)
    pts_b = uniform_points(10000, 12.0, seed=5)
    ro, rd, _ = synth_rays(1, 32, 32, seed=3)
    with torch.no_grad():
        idx = model._calculate_global_distances(ro, rd, pts_b)
    out.update(b_idx=np.sort(npf(idx), axis=-1).astype(np.int32), b_points_sum=stats(pts_b))
    # (c) two images (two origins), 8x8 rays each, lattice cloud
    ro, rd, _ = synth_rays(2, 8, 8, seed=7)
    with torch.no_grad():
        idx = model._calculate_global_distances(ro, rd, pts)
    out.update(c_idx=np.sort(npf(idx), axis=-1).astype(np.int32))
    # (d) un-normalised directions (the selection step must not renormalise)
    ro, rd, _ = synth_rays(1, 8, 8, seed=9)
    rd = rd * 1.7
    with torch.no_grad():
        idx = model._calculate_global_distances(ro, rd, pts)
        _, _, proj, D = model._calculate_distances(ro, rd, pts[idx], None)
    out.update(d_idx=np.sort(npf(idx), axis=-1).astype(np.int32), d_proj=npf(proj), d_D=npf(D),
               d_idx_raw=npf(idx).astype(np.int32))
    save("g34_knn_geometry.npz", **out)


# ----------------------------------------------------------------------------------- G5-G7
def model_case(tag, cfg, n_img=1, hw=16, ray_seed=0, keep_rows=64):
    model = build(cfg)
    ro, rd, c2w = synth_rays(n_img, hw, hw, seed=ray_seed)
    out = {"points": npf(model.points)}
    with torch.no_grad():
        fused, attn = model.evaluate(ro, rd, c2w)
        idx = model.select_k_ind.clone()
    # intermediates of the attention block on the reference's own neighbour order
    pts_sel, _ = model._get_points(ro, rd, c2w)
    key, query, value, kx, qx, vx = model._get_kqv(ro, rd, pts_sel, c2w, idx)
    with torch.no_grad():
        k, q, v, scores = model.proximity_attn(key, query, value, kx, qx, vx)
    kk = idx.shape[-1]
    out.update(idx_raw=npf(idx).astype(np.int32),
               fused=npf(fused.squeeze(-2)), attn=npf(attn.squeeze(-1)),
               K_head=npf(k[:8]), Q_head=npf(q[:32, 0]), V_head=npf(v[:keep_rows]),
               scores=npf(scores.reshape(-1, kk)),
               K_stats=stats(k), Q_stats=stats(q), V_stats=stats(v))
    # forward + gradients of mean((rgb-0.5)^2)
    model.clear_grad()
    rgb = model(ro, rd, c2w)
    loss = torch.mean((rgb - 0.5) ** 2)
    loss.backward()
    out.update(rgb=npf(rgb), loss=np.array(loss.item()))
    full = ("points", "points_influ_scores", "pc_feats", "bkg_feats",
            "proximity_attn.embed.embed_k.innorm.a_2", "proximity_attn.embed.embed_k.innorm.b_2",
            "proximity_attn.embed.embed_k.outnorm.a_2", "proximity_attn.embed.embed_q.outnorm.b_2",
            "proximity_attn.embed.embed_k.mlp.model.1.weight", "proximity_attn.embed.embed_v.mlp.model.1.weight",
            "proximity_attn.embed.embed_q.mlp.model.1.weight", "proximity_attn.embed.embed_v.mlp.model.11.weight",
            "proximity_attn.attention_layer.w_q.bias", "proximity_attn.attention_layer.w_k.bias")
    names, gstats = [], []
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        names.append(name)
        gstats.append(stats(p.grad))
        if name in full or p.dim() == 1:
            out["grad/" + name] = npf(p.grad)
    out["grad_names"] = np.array(names)
    out["grad_stats"] = np.stack(gstats)
    save("g567_%s.npz" % tag, **out)
    return model


def g7_trajectory(cfg):
    """Three reference train_step calls (models/model.py:439-460 + train.py:155-179)."""
    model = build(cfg)
    ro, rd, c2w = synth_rays(1, 16, 16, seed=0)
    g = torch.Generator().manual_seed(21)
    tgt = torch.rand((1, 16, 16, 3), generator=g)

    class DS:
        def get_c2w(self, i):
            return c2w[0]

    loss_fn = get_loss(cfg["training"]["losses"])
    args = DictAsMember(copy.deepcopy(cfg))
    losses = []
    for step in range(3):
        loss, _ = ref_train.train_step(step + 1, model, "cpu", DS(), ([0], None, tgt, rd, ro), loss_fn, args)
        losses.append(loss)
    sd = model.state_dict()
    names = [n for n in sd if sd[n].is_floating_point()]
    save("g7_trajectory.npz", losses=np.array(losses, dtype=np.float64), target=npf(tgt),
         points_after=npf(sd["points"]), influ_after=npf(sd["points_influ_scores"]),
         names=np.array(names), stats_after=np.stack([stats(sd[n]) for n in names]),
         attn_lr=np.array(model.attn_lr), pts_lr=np.array(model.pts_lr))
    print("trajectory", losses)


def g8_manifest(model):
    sd = model.state_dict()
    man = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()}
    man["__optimizers__"] = sorted(model.optimizers.keys())
    n_attn = sum(p.numel() for p in model.proximity_attn.parameters())
    n_rend = sum(p.numel() for p in model.renderer.parameters())
    man["__counts__"] = {"attn": n_attn, "renderer": n_rend}
    json.dump(man, open(os.path.join(HERE, "g8_manifest.json"), "w"), indent=0, sort_keys=True)
    print("wrote g8_manifest.json", n_attn, n_rend)


def g8_init(cfg, name):
    """Statistics of the reference's freshly initialised parameters under setup_seed(1)."""
    setup_seed(1)
    model = get_model(DictAsMember(copy.deepcopy(cfg)), "cpu")
    sd = model.state_dict()
    names = [n for n in sd if sd[n].is_floating_point()]
    save(name, names=np.array(names), stats=np.stack([stats(sd[n]) for n in names]),
         points=npf(sd["points"]), pc_feats_head=npf(sd["pc_feats"][:8]),
         wk_head=npf(sd["proximity_attn.attention_layer.w_k.weight"][:4]),
         inc_bias=npf(sd["renderer.inc.double_conv.0.bias"]))


def g9_dp(cfg):
    """2-image batch == mean of the two single-image gradients (the 2-rank DP oracle)."""
    model = build(cfg)
    ro, rd, c2w = synth_rays(2, 16, 16, seed=13)
    g = torch.Generator().manual_seed(22)
    tgt = torch.rand((2, 16, 16, 3), generator=g)
    out = {"target": npf(tgt)}

    def grads(sl):
        model.clear_grad()
        for p in model.parameters():
            p.grad = None
        rgb = model(ro[sl], rd[sl], c2w[sl])
        loss = torch.mean((rgb - tgt[sl]) ** 2)
        loss.backward()
        return loss.item(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    for tag, sl in (("both", slice(0, 2)), ("img0", slice(0, 1)), ("img1", slice(1, 2))):
        loss, gr = grads(sl)
        out[tag + "/loss"] = np.array(loss)
        out[tag + "/points"] = npf(gr["points"])
        out[tag + "/influ"] = npf(gr["points_influ_scores"])
        out[tag + "/wq_bias"] = npf(gr["proximity_attn.attention_layer.w_q.bias"])
        out[tag + "/outc_bias"] = npf(gr["renderer.outc.conv.bias"])
        names = sorted(gr)
        out[tag + "/names"] = np.array(names)
        out[tag + "/stats"] = np.stack([stats(gr[n]) for n in names])
    save("g9_dp.npz", **out)


VARIANTS = {"geoms": {"points": {"init_num": 1000}, "point_feats": {"use_ink": True, "use_inv": True}},
            "models": {"normalize_topk_attn": False, "attn": {"embed": {"embed_type": 2}}}}


def ray_samples(rd):
    """A compact witness of a full (N,H,W,3) ray map: border rows / columns, a centre block, float64 sums."""
    N, H, W, _ = rd.shape
    return dict(rows=npf(rd[:, [0, 1, H // 2, H - 1]]), cols=npf(rd[:, :, [0, 1, W // 2, W - 1]]),
                block=npf(rd[:, H // 2 - 16:H // 2 + 16, W // 3:W // 3 + 32]), sums=np.stack([stats(rd[n]) for n in range(N)]))


def g10_rays():
    """Reference get_rays (dataset/utils.py:81-96) and extract_patches (:99-118) for the Blender 800 x 800 camera and a
    Tanks&Temples-style 1088 x 640 camera with fx != fy (dataset/load_t2.py:63-76 after factor 2)."""
    import math
    from dataset.utils import get_rays as ref_get_rays, extract_patches as ref_extract
    from papr_amd.data import make_cameras
    out = {}
    cams = {"blender": (800, 800, 0.5 * 800 / math.tan(0.5 * 0.6911112070083618), 0.5 * 800 / math.tan(0.5 * 0.6911112070083618), 10.0),
            "t2": (640, 1088, 581.7877, 583.2061, 30.0)}
    for tag, (H, W, fx, fy, scale) in cams.items():
        c2w = make_cameras(2, seed=4 if tag == "blender" else 9, coord_scale=scale)
        ro, rd = ref_get_rays(H, W, fx, fy, c2w)
        out[tag + "/cam"] = np.array([H, W, fx, fy, scale], dtype=np.float64)
        out[tag + "/c2w"] = npf(c2w)
        out[tag + "/rays_o"] = npf(ro)
        for k, v in ray_samples(rd).items():
            out[tag + "/" + k] = v
        # extract_patches under np.random.seed(7): the crop offsets come from the global numpy stream
        ph, pw = (160, 160) if tag == "blender" else (180, 180)
        args = DictAsMember({"patches": {"height": ph, "width": pw, "max_patches": 2}})
        imgs = torch.arange(2 * H * W, dtype=torch.float32).reshape(2, H, W, 1)         # pixel id as "image": reveals the offsets
        np.random.seed(7)
        img_p, rayd_p, rayo_p, n = ref_extract(imgs, ro, rd, args)
        off = img_p[:, :, 0, 0, 0].astype(np.int64) % (H * W)
        out[tag + "/patch_hw"] = np.stack([off // W, off % W], -1)                       # (N, n_patches, 2)
        out[tag + "/patch_rayd_sums"] = np.stack([[stats(torch.from_numpy(rayd_p[i, j])) for j in range(n)] for i in range(2)])
        out[tag + "/patch_rayd_corner"] = rayd_p[:, :, :4, :4]
    save("g10_rays.npz", **out)


def g11_t2():
    """Reference load_meta_data / RINDataset on a generated Tanks&Temples-format scene."""
    import tempfile
    from dataset.utils import load_meta_data
    from dataset.dataset import RINDataset
    with tempfile.TemporaryDirectory() as base:
        write_t2_fixture(base)
        out = {}
        for mode in ("train", "test"):
            args = DictAsMember({"type": "t2", "path": base, "factor": 1, "read_offline": True, "white_bg": False, "coord_scale": 30.0,
                                 "extract_patch": False, "extract_online": False, "patches": {"height": 8, "width": 8, "max_patches": 1}})
            ds = RINDataset(args, mode=mode)
            out[mode + "/hwf"] = np.array([ds.H, ds.W, ds.focal_x, ds.focal_y], dtype=np.float64)
            out[mode + "/c2w"] = npf(ds.c2w)
            out[mode + "/images_sums"] = np.stack([stats(ds.images[i]) for i in range(ds.images.shape[0])])
            out[mode + "/image0_head"] = npf(ds.images[0, :6, :8])
            out[mode + "/rayd0"] = npf(ds.rayd[0])
            out[mode + "/rayo"] = npf(ds.rayo)
        save("g11_t2.npz", **out)


def g12_lpips():
    """Reference LPNet (models/lpips.py:86-125) with a formula-filled backbone and heads (the ImageNet weights cannot be
    fetched; the ARITHMETIC is what is pinned): loss value and its gradient for a seeded image pair."""
    from papr_amd.lpips import vgg16_features
    import models.lpips as ref_lpips
    tv = sys.modules["torchvision.models"]

    class _Holder:
        def __init__(self):
            self.features = vgg16_features()
    tv.vgg16 = lambda weights=None: _Holder()
    tv.VGG16_Weights = types.SimpleNamespace(IMAGENET1K_V1=None)
    ref_lpips.tv = tv
    cwd = os.getcwd()
    os.chdir(REF)                                    # the reference loads ./vgg.pth (its own data file) -- overwritten just below
    try:
        net = ref_lpips.LPNet()
    finally:
        os.chdir(cwd)
    sd = {}
    for sl in (net.net.slice1, net.net.slice2, net.net.slice3, net.net.slice4, net.net.slice5):
        for idx, layer in sl.named_children():
            for k, v in layer.state_dict().items():
                sd["features.%s.%s" % (idx, k)] = v
    formula_fill(sd, salt=3)
    with torch.no_grad():
        for i, lin in enumerate(net.lins):
            lin.weight.copy_(torch.rand(lin.weight.shape, generator=torch.Generator().manual_seed(40 + i)))
    g = torch.Generator().manual_seed(31)
    a = torch.rand((2, 40, 48, 3), generator=g).requires_grad_(True)
    b = torch.rand((2, 40, 48, 3), generator=g)
    val = net(a, b)
    val.backward()
    one = net(a[:1].detach(), b[:1])
    save("g12_lpips.npz", a=npf(a), b=npf(b), value=np.array(val.item()), value_first=np.array(one.item()), grad_a=npf(a.grad),
         lin_sums=np.array([float(l.weight.sum()) for l in net.lins]))


# ----------------------------------------------------------------------------------- G13 (round 3)
G13_CFG = {"geoms": {"points": {"init_num": 400, "select_k": 12}, "point_feats": {"dim": 16}},
           "models": {"use_renderer": False, "attn": {"d_model": 64, "embed": {
               "k_L": [4, 4, 4], "q_L": [4], "v_L": [4, 4],
               "key": {"d_ff": 64, "d_ff_out": 64, "n_ff_layer": 3},
               "query": {"d_ff": 64, "d_ff_out": 64, "n_ff_layer": 3},
               "value": {"d_ff": 64, "d_ff_out": 3, "n_ff_layer": 4}}}}}


def g13_ref_checkpoint():
    """A checkpoint directory written by the REFERENCE's own PAPR.save (models/model.py:562-586) after three of its own train
    steps from a random (seeded) initialisation, and what the reference renders from it: the files a user of the reference
    has on disk (model.pth = {str(step): state_dict}, optimizers.pth, schedulers.pth, scaler.pth)."""
    cfg = load_cfg("nerfsyn/chair.yml", **G13_CFG)
    setup_seed(3)                                       # (seed 1 happens to give q.k < 0 for every pair of this tiny model: all scores 0 behind the ReLU; seed 3: 44 % positive)
    model = get_model(DictAsMember(copy.deepcopy(cfg)), "cpu")
    with torch.no_grad():                               # untrained influence scores are exactly zero: give the attention something to weigh
        model.points_influ_scores.uniform_(0.0, 1.0, generator=torch.Generator().manual_seed(7))
    ro, rd, c2w = synth_rays(2, 8, 8, seed=11)
    tgt = torch.rand(2, 8, 8, 3, generator=torch.Generator().manual_seed(12))
    loss_fn = get_loss(cfg["training"]["losses"])

    class DS:                                           # what train_step asks of the dataset (train.py:158-161)
        def get_c2w(self, idx): return c2w[idx]

    args = DictAsMember(copy.deepcopy(cfg))
    for step in range(3):
        ref_train.train_step(step, model, "cpu", DS(), (torch.tensor([0, 1]), None, tgt, rd, ro), loss_fn, args)
    out_dir = os.path.join(HERE, "g13_ref_ckpt")
    os.makedirs(out_dir, exist_ok=True)
    model.save(3, out_dir)
    with torch.no_grad():
        fused, attn = model.evaluate(ro, rd, c2w)
        rgb = model(ro, rd, c2w)
    save("g13_ref_ckpt_outputs.npz", rays_o=npf(ro), rays_d=npf(rd), c2w=npf(c2w), fused=npf(fused.squeeze(-2)), attn=npf(attn.squeeze(-1)),
         rgb=npf(rgb), idx=npf(model.select_k_ind).astype(np.int32), points=npf(model.points))
    for f in sorted(os.listdir(out_dir)):
        print("wrote g13_ref_ckpt/%s %.1f KB" % (f, os.path.getsize(os.path.join(out_dir, f)) / 1024))


# ----------------------------------------------------------------------------------- G14
G14_COMB = ["mean", "random", "random-softmax", "weighted", "duplicate"]
G14_SAMPLE = ["random", "top-knn-std", "top-knn-mean", "top-knn-max", "top-knn-min", "influ-scores-max", "influ-scores-min"]


def g14_cloud(P, seed, ties=False):
    """Seeded cloud of uniform points; ties: plus a lattice patch (equal neighbour distances: ties in the site ranking AND in the neighbour
    order, which scipy's KDTree resolves by its traversal) and a few exact duplicates."""
    g = torch.Generator().manual_seed(seed)
    pts = (torch.rand((P, 3), generator=g) * 2 - 1) * 12.0
    if ties:
        n_lat = min(64, P // 4)
        lat = torch.stack(torch.meshgrid(*[torch.arange(4.0)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n_lat] * 1.5 - 3.0
        pts[:n_lat] = lat
        pts[P - 5:] = pts[20:25]                                   # five duplicated points
    influ = torch.rand((P, 1), generator=g)
    feats = torch.randn((P, 8), generator=g)
    return pts.float(), influ.float(), feats.float()


def g14_grow():
    """The reference's growth / pruning procedures (models/utils.py:9-109 add_points_knn; models/model.py:335-394 PAPR.prune_points / add_points) under
    fixed numpy seeds: every (comb_type, sample_type) pair, the `N <= add_num` branches, and the model-level calls incl. the `max_points` guard."""
    from models.utils import add_points_knn
    out = {}
    pts, influ, feats = g14_cloud(600, 14)
    out["cloud/points"], out["cloud/influ"], out["cloud/feats"] = npf(pts), npf(influ), npf(feats)
    case = 0
    for comb in G14_COMB:
        for samp in G14_SAMPLE:
            np.random.seed(1400 + case)
            nc, n, ni, nf = add_points_knn(pts.clone(), influ.clone(), add_num=37, k=3, comb_type=comb, sample_type=samp, sample_k=10, point_features=feats.clone())
            tag = "pair/%s/%s" % (comb, samp)
            out[tag + "/coords"], out[tag + "/influ"], out[tag + "/feats"] = np.asarray(nc, dtype=np.float32), np.asarray(ni, dtype=np.float32), np.asarray(nf, dtype=np.float32)
            out[tag + "/seed_n"] = np.array([1400 + case, n])
            assert n == 37
            case += 1
    # a cloud with exact ties (lattice patch, duplicated points): the host procedure uses the same KDTree as the reference and must follow it
    # through them; a brute-force search cannot promise the KDTree's order among equal distances, so the device test stays on the cloud above
    tpts, tinflu, tfeats = g14_cloud(600, 14, ties=True)
    out["ties/points"], out["ties/influ"], out["ties/feats"] = npf(tpts), npf(tinflu), npf(tfeats)
    for i, (comb, samp) in enumerate((("mean", "top-knn-max"), ("random", "top-knn-min"), ("weighted", "top-knn-std"), ("random-softmax", "random"))):
        np.random.seed(1490 + i)
        nc, n, ni, nf = add_points_knn(tpts.clone(), tinflu.clone(), add_num=37, k=3, comb_type=comb, sample_type=samp, sample_k=10, point_features=tfeats.clone())
        tag = "ties/%s/%s" % (comb, samp)
        out[tag + "/coords"], out[tag + "/influ"], out[tag + "/feats"] = np.asarray(nc, dtype=np.float32), np.asarray(ni, dtype=np.float32), np.asarray(nf, dtype=np.float32)
        out[tag + "/seed_n"] = np.array([1490 + i, n])
    # the config's own pair at other k / sample_k, and without features
    np.random.seed(1450)
    nc, n, ni, nf = add_points_knn(pts.clone(), influ.clone(), add_num=50, k=5, comb_type="random", sample_type="top-knn-std", sample_k=6, point_features=None)
    out["k5/coords"], out["k5/influ"] = np.asarray(nc, dtype=np.float32), np.asarray(ni, dtype=np.float32)
    assert nf is None
    # N <= add_num: random comb types draw sites with replacement, the others take every point once
    spts, sinflu, sfeats = g14_cloud(20, 15)
    out["small/points"], out["small/influ"], out["small/feats"] = npf(spts), npf(sinflu), npf(sfeats)
    for i, comb in enumerate(G14_COMB):
        np.random.seed(1460 + i)
        nc, n, ni, nf = add_points_knn(spts.clone(), sinflu.clone(), add_num=30, k=3, comb_type=comb, sample_type="top-knn-std", sample_k=10, point_features=sfeats.clone())
        tag = "small/%s" % comb
        out[tag + "/coords"], out[tag + "/influ"], out[tag + "/feats"] = np.asarray(nc, dtype=np.float32), np.asarray(ni, dtype=np.float32), np.asarray(nf, dtype=np.float32)
        out[tag + "/seed_n"] = np.array([1460 + i, n])
    # model level: PAPR.prune_points (both prune types) and PAPR.add_points (default config pair; with and without the `max_points` guard).
    # The initial state is the seeded construction itself (setup_seed(1); the build reproduces that stream bit for bit, test_initialisation_...),
    # so only the influence scores set here, the per-point results and -- for the 64-wide features -- a float64 row checksum plus the NEW rows are stored.
    rowsum = lambda t: t.detach().double().sum(1).numpy()
    for ptype, tag in (("<", "lt"), (">", "gt")):
        cfg = load_cfg("nerfsyn/chair.yml", geoms={"points": {"init_num": 1000}}, training={"prune_type": ptype})
        setup_seed(1)
        model = get_model(DictAsMember(copy.deepcopy(cfg)), "cpu")
        with torch.no_grad():
            model.points_influ_scores.copy_(torch.rand((1000, 1), generator=torch.Generator().manual_seed(16)))
            model.points_influ_scores[::7] = 0.0                     # never-selected points keep exactly 0.0: `0 > 0` is false, they go
            # the 10 x 10 x 10 initial lattice is all ties; a trained cloud has none: seeded jitter (the device search cannot follow the KDTree's tie order)
            model.points.add_(0.3 * torch.randn((1000, 3), generator=torch.Generator().manual_seed(18)))
        if tag == "lt":
            out["model/influ0"] = npf(model.points_influ_scores)
            out["model/points0_sum"], out["model/feats0_rowsum"] = npf(model.points).astype(np.float64).sum(0), rowsum(model.pc_feats)
        n_drop = model.prune_points(0.3)
        out["model/%s/n_drop" % tag] = np.array(int(n_drop))
        out["model/%s/points" % tag], out["model/%s/influ" % tag], out["model/%s/feats_rowsum" % tag] = npf(model.points), npf(model.points_influ_scores), rowsum(model.pc_feats)
        if tag == "lt":
            P1 = model.points.shape[0]
            np.random.seed(1470)
            n_add = model.add_points(100)
            out["model/add/n"] = np.array(int(n_add))
            out["model/add/points"], out["model/add/influ"] = npf(model.points), npf(model.points_influ_scores)
            out["model/add/feats_rowsum"], out["model/add/new_feats"] = rowsum(model.pc_feats), npf(model.pc_feats[P1:])
            assert model.points.shape[0] == P1 + 100
    for tag, max_points in (("cap", 650), ("full", 500)):
        cfg = load_cfg("nerfsyn/chair.yml", geoms={"points": {"init_num": 600}}, max_points=max_points)
        setup_seed(1)
        model = get_model(DictAsMember(copy.deepcopy(cfg)), "cpu")
        with torch.no_grad():
            model.points_influ_scores.copy_(torch.rand((600, 1), generator=torch.Generator().manual_seed(17)))
            model.points.add_(0.3 * torch.randn((600, 3), generator=torch.Generator().manual_seed(19)))
        out["model/%s/influ0" % tag] = npf(model.points_influ_scores)
        np.random.seed(1480)
        n_add = model.add_points(100)
        out["model/%s/n" % tag] = np.array(int(n_add))
        out["model/%s/points" % tag], out["model/%s/influ" % tag] = npf(model.points), npf(model.points_influ_scores)
        out["model/%s/feats_rowsum" % tag], out["model/%s/new_feats" % tag] = rowsum(model.pc_feats), npf(model.pc_feats[600:])
        print("max_points", max_points, "->", int(n_add), tuple(model.points.shape))
    save("g14_grow.npz", **out)


# ----------------------------------------------------------------------------------- G15
G15_OVER = {
    "seed": 4, "geoms": {"points": {"init_num": 1000}},
    "dataset": {"patches": {"height": 16, "width": 16}},
    "eval": {"step": 100, "img_idx": 1, "max_height": 20, "max_width": 20, "save_fig": False},
    # 360 steps with everything the long schedule has: warm-up (60 steps), pruning every 60 steps from step 120, growth every 120 from 240;
    # lr_factor 0.3: at full rate this tiny problem overshoots behind the warm-up (eval PSNR 13 -> 8 dB), at 0.3 it improves monotonically
    "training": {"steps": 360, "prune_steps": 60, "prune_start": 120, "prune_stop": 330, "prune_thresh": 0.0,
                 "add_steps": 120, "add_start": 240, "add_stop": 360, "add_num": 80,
                 "lr": {"lr_factor": 0.3, "attn": {"warmup": 60}, "points_influ_scores": {"warmup": 60}, "feats": {"warmup": 60}, "generator": {"warmup": 60}}},
}


def g15_run(threads):
    """One run of the reference's loop; returns the per-step / per-event record."""
    import tempfile
    torch.set_num_threads(threads)
    with tempfile.TemporaryDirectory() as tmp:
        scene = os.path.join(tmp, "scene") + "/"
        write_blender_fixture(scene)
        over = copy.deepcopy(G15_OVER)
        over["dataset"]["path"] = scene
        over["eval"]["dataset"] = {"path": scene}
        over["save_dir"] = os.path.join(tmp, "exp")
        over["index"] = "g15"
        cfg = load_cfg("nerfsyn/chair.yml", **over)
        ecfg = copy.deepcopy(cfg)
        ecfg["dataset"].update(ecfg["eval"]["dataset"])            # (train.py:351-353)
        args, eargs = DictAsMember(copy.deepcopy(cfg)), DictAsMember(ecfg)
        os.makedirs(os.path.join(cfg["save_dir"], cfg["index"]), exist_ok=True)
        from dataset import get_dataset as ref_get_dataset
        setup_seed(cfg["seed"])
        model = get_model(args, "cpu")                              # main(): train.py:302-307
        dataset = ref_get_dataset(args.dataset, mode="train")
        eval_dataset = ref_get_dataset(eargs.dataset, mode="test")
        rec = {"loss": [], "img": [], "P": [], "influ_pos": [], "tgt_sum": []}
        orig_step = ref_train.train_step

        def spy(step, model, device, dataset, batch, loss_fn, a):   # per-step record around the reference's own train_step
            rec["img"].append(int(batch[0][0]))
            rec["tgt_sum"].append(float(batch[2].double().sum()))     # which crop: the checksum of the target patch
            rec["P"].append(int(model.points.shape[0]))
            out = orig_step(step, model, device, dataset, batch, loss_fn, a)
            rec["loss"].append(float(out[0]))
            rec["influ_pos"].append(int((model.points_influ_scores > 0).sum()))
            return out
        ref_train.train_step = spy
        ev_log = []                                                 # (step, kind 0 = prune / 1 = add, count, points after)
        for kind, name in ((0, "prune_points"), (1, "add_points")):
            def wrap(fn, kind):
                def call(*a, **k):
                    n = fn(*a, **k)
                    ev_log.append((len(rec["loss"]), kind, int(n), int(model.points.shape[0])))
                    return n
                return call
            setattr(model, name, wrap(getattr(model, name), kind))
        losses = [[], [], []]
        try:
            ref_train.train_and_eval(0, model, "cpu", dataset, eval_dataset, losses, args)
        finally:
            ref_train.train_step = orig_step
        sd = model.state_dict()
        print("threads", threads, "events (step, prune 0 / add 1, count, points after):", ev_log)
        print("eval psnrs:", losses[2], "train-loss averages:", losses[0])
        return dict(loss=np.array(rec["loss"]), tgt_sum=np.array(rec["tgt_sum"]), img=np.array(rec["img"], dtype=np.int32), P=np.array(rec["P"], dtype=np.int32),
                    influ_pos=np.array(rec["influ_pos"], dtype=np.int32), events=np.array(ev_log, dtype=np.int32),
                    eval_psnrs=np.array(losses[2]), eval_losses=np.array(losses[1]), train_loss_avgs=np.array(losses[0]),
                    points_final=npf(sd["points"]), influ_final=npf(sd["points_influ_scores"]),
                    attn_lr=np.array(model.attn_lr), pts_lr=np.array(model.pts_lr))


def g15_dynamics():
    """The reference's own training loop -- train.train_and_eval (train.py:182-300) with its DataLoader, train_step, prune / add schedule,
    init_optimizers(step) re-creation and eval_step -- on CPU for 360 steps of a tiny nerf_synthetic-format directory (formula.write_blender_fixture).
    Run twice with different thread counts (torch's CPU reductions change their summation order with it): run `a` is the pin, `b` shows how far
    the REFERENCE drifts from itself over these steps -- the yardstick for the band the build is held to."""
    a, b = g15_run(8), g15_run(3)
    assert np.array_equal(a["img"], b["img"]) and np.array_equal(a["tgt_sum"], b["tgt_sum"])
    out = {"a/" + k: v for k, v in a.items()}
    out.update({"b/" + k: v for k, v in b.items() if k not in ("img", "tgt_sum", "points_final", "influ_final")})
    save("g15_dynamics.npz", cfg_json=np.array(json.dumps(G15_OVER)), **out)


# ----------------------------------------------------------------------------------- G16
G16_ACTS = ["none", "relu", "leakyrelu", "+1", "relu+1", "tanh", "shifted_tanh", "sigmoid", "gelu", "gaussian", "quadratic", "multi-quadratic",
            "laplacian", "super-gaussian", "expsin", "clamp", "sine", "softplus_1.5_2.0_-0.25"]


def g16_last_act():
    """models.last_act: the reference's activation_func (models/utils.py:183-229) with its default arguments on one seeded tensor, and the
    state-dict keys each choice adds to the model (`last_act.a`, ...)."""
    from models.utils import activation_func
    x = (torch.randn((2, 5, 7, 3), generator=torch.Generator().manual_seed(16)) * 1.5)
    out = {"x": npf(x)}
    for name in G16_ACTS:
        layer = activation_func(name)
        out["y/" + name] = npf(layer(x.clone()))
        out["keys/" + name] = np.array(sorted(layer.state_dict().keys()))
    save("g16_last_act.npz", **out)


# ----------------------------------------------------------------------------------- G17
def _rel(a, b):
    """(L-inf, rms) of a - b over the L-inf of b (float64)."""
    a, b = a.detach().double(), b.detach().double()
    s = max(b.abs().max().item(), 1e-30)
    return np.array([(a - b).abs().max().item() / s, (a - b).pow(2).mean().sqrt().item() / s])


def g17_amp(tag, scene):
    """The reference with `use_amp: true` as every shipped YAML has it (configs/default.yml:6-7): attention block and U-Net under fp16 autocast
    (shim 4 above), GradScaler live.  Same seeded model, rays and neighbour order as g567_<tag>: evaluate / forward outputs, intermediates of the
    attention block, gradients of mean((rgb - 0.5)^2) taken the way train_step takes them (scaler.scale(loss).backward(), divided by the scale
    here), and the reference's OWN distance between its AMP and its fp32 results -- the yardstick the build's use_amp path is held to."""
    small = {"geoms": {"points": {"init_num": 1000}}}
    cfg32 = load_cfg(scene, **small)
    cfg16 = copy.deepcopy(cfg32)
    cfg16["use_amp"] = True
    assert cfg16["amp_dtype"] == "float16"
    m32, m16 = build(cfg32), build(cfg16)
    assert m16.scaler.is_enabled() and m16.scaler.get_scale() == 65536.0
    ro, rd, c2w = synth_rays(1, 16, 16, seed=0)
    out, yard = {}, {}
    with torch.no_grad():
        f32_, a32_ = m32.evaluate(ro, rd, c2w)
        idx32 = m32.select_k_ind.clone()
        f16_, a16_ = m16.evaluate(ro, rd, c2w)
        idx = m16.select_k_ind.clone()
    assert torch.equal(idx, idx32)                  # (the selection runs outside autocast)
    kk = idx.shape[-1]

    def block(m):
        pts_sel, _ = m._get_points(ro, rd, c2w)
        key, query, value, kx, qx, vx = m._get_kqv(ro, rd, pts_sel, c2w, idx)
        with torch.no_grad():
            return m.proximity_attn(key, query, value, kx, qx, vx)
    k32, q32, v32, s32 = block(m32)
    k16, q16, v16, s16 = block(m16)
    out.update(idx_raw=npf(idx).astype(np.int32), fused=npf(f16_.squeeze(-2).float()), attn=npf(a16_.squeeze(-1).float()),
               K_head=npf(k16[:8].float()), Q_head=npf(q16[:32, 0].float()), V_head=npf(v16[:64].float()), scores=npf(s16.reshape(-1, kk).float()),
               dtypes=np.array([str(t.dtype) for t in (k16, q16, v16, s16, f16_, a16_)]))
    yard.update(fused=_rel(f16_, f32_), attn=_rel(a16_, a32_), K=_rel(k16, k32), Q=_rel(q16, q32), V=_rel(v16, v32), scores=_rel(s16, s32))

    def grads(m):
        m.clear_grad()
        rgb = m(ro, rd, c2w)
        loss = torch.mean((rgb - 0.5) ** 2)
        m.scaler.scale(loss).backward()
        sc = m.scaler.get_scale() if m.scaler.is_enabled() else 1.0
        return rgb.detach(), loss.item(), {n: p.grad.detach() / sc for n, p in m.named_parameters() if p.grad is not None}
    rgb32, loss32, g32 = grads(m32)
    rgb16, loss16, g16 = grads(m16)
    assert rgb16.dtype == torch.float32 and all(torch.isfinite(v).all() for v in g16.values())
    out.update(rgb=npf(rgb16), loss=np.array(loss16), loss_fp32=np.array(loss32))
    yard["rgb"] = _rel(rgb16, rgb32)
    full = ("points", "points_influ_scores", "pc_feats",
            "proximity_attn.embed.embed_k.mlp.model.1.weight", "proximity_attn.embed.embed_v.mlp.model.1.weight",
            "proximity_attn.embed.embed_q.mlp.model.1.weight", "proximity_attn.embed.embed_v.mlp.model.11.weight",
            "proximity_attn.attention_layer.w_q.bias", "proximity_attn.attention_layer.w_k.bias", "renderer.outc.conv.bias",
            "renderer.inc.double_conv.0.bias")
    names = sorted(g16)
    assert names == sorted(g32)
    for n in names:
        if n in full or g16[n].dim() == 1:
            out["grad/" + n] = npf(g16[n])
    out["grad_names"] = np.array(names)
    out["grad_stats"] = np.stack([stats(g16[n]) for n in names])
    out["grad_yard"] = np.stack([_rel(g16[n], g32[n]) for n in names])       # per tensor: (L-inf, rms) of AMP - fp32 over max |fp32 gradient|
    for k_, v_ in yard.items():
        out["yard/" + k_] = v_
    print(tag, "reference AMP vs its own fp32 (L-inf, rms relative to the fp32 tensor's max):", {k_: v_.round(6).tolist() for k_, v_ in yard.items()})
    print(tag, "gradients: worst rms %.3e, worst L-inf %.3e" % (out["grad_yard"][:, 1].max(), out["grad_yard"][:, 0].max()))

    # three train_step calls of the reference (train.py:155-179) with the flag on: the G7 case
    model = build(cfg16)
    g = torch.Generator().manual_seed(21)
    tgt = torch.rand((1, 16, 16, 3), generator=g)

    class DS:
        def get_c2w(self, i):
            return c2w[0]

    loss_fn = get_loss(cfg16["training"]["losses"])
    args = DictAsMember(copy.deepcopy(cfg16))
    losses, scales = [], []
    for step in range(3):
        loss, _ = ref_train.train_step(step + 1, model, "cpu", DS(), ([0], None, tgt, rd, ro), loss_fn, args)
        losses.append(loss)
        scales.append(model.scaler.get_scale())
    sd = model.state_dict()
    out.update(traj_losses=np.array(losses, dtype=np.float64), traj_scales=np.array(scales), traj_target=npf(tgt),
               traj_points_after=npf(sd["points"]), traj_influ_after=npf(sd["points_influ_scores"]))
    print(tag, "AMP trajectory", losses, "scales", scales)
    save("g17_amp_%s.npz" % tag, **out)


if __name__ == "__main__":
    small = {"geoms": {"points": {"init_num": 1000}}}
    if "--amp" in sys.argv:                          # G17: the reference under `use_amp: true` (CPU autocast, shim 4)
        import warnings
        warnings.simplefilter("ignore")
        g17_amp("chair1k", "nerfsyn/chair.yml")
        g17_amp("lego1k", "nerfsyn/lego.yml")
        sys.exit(0)
    if "--g17" in sys.argv:                          # weight-normalised embedding MLPs (`use_wn: true`, models/mlp.py:21,35-36): a tiny model
        wn = copy.deepcopy(G13_CFG)
        wn["geoms"]["points"]["init_num"] = 1000
        for k in ("key", "query", "value"):
            wn["models"]["attn"]["embed"][k]["use_wn"] = True
        import warnings
        warnings.simplefilter("ignore")
        model_case("wn_tiny", load_cfg("nerfsyn/chair.yml", **wn), n_img=2, hw=8, ray_seed=4)
        sys.exit(0)
    if "--g18" in sys.argv:                          # half layers (`half_layers`, models/mlp.py:27-30; no shipped scene file sets them): a tiny model
        hc = copy.deepcopy(G13_CFG)
        hc["geoms"]["points"]["init_num"] = 1000
        hc["models"]["attn"]["embed"]["key"]["half_layers"] = [1]
        hc["models"]["attn"]["embed"]["value"]["half_layers"] = [2]
        model_case("half_tiny", load_cfg("nerfsyn/chair.yml", **hc), n_img=2, hw=8, ray_seed=4)
        sys.exit(0)
    if "--g16" in sys.argv:
        g16_last_act()
        sys.exit(0)
    if "--g14" in sys.argv:
        g14_grow()
        sys.exit(0)
    if "--g15" in sys.argv:
        sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
        g15_dynamics()
        sys.exit(0)
    if "--round4" in sys.argv:
        sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
        g14_grow()
        g15_dynamics()
        g16_last_act()
        sys.exit(0)
    if "--round3" in sys.argv:
        g13_ref_checkpoint()
        sys.exit(0)
    if "--lpips" in sys.argv:
        sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
        g12_lpips()
        sys.exit(0)
    if "--round2" in sys.argv:                       # fixtures added in round 2 (the round-1 files are left untouched)
        sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
        model_case("variants1k", load_cfg("nerfsyn/chair.yml", **VARIANTS))
        g10_rays()
        g11_t2()
        g12_lpips()
        sys.exit(0)
    cfg1 = load_cfg("nerfsyn/chair.yml", **small)
    if "--init-only" in sys.argv:
        g8_init(cfg1, "g8_init_chair1k.npz")
        g8_init(load_cfg("nerfsyn/lego.yml", **small), "g8_init_lego1k.npz")
        sys.exit(0)
    g1_g2()
    m = model_case("chair1k", cfg1)
    g3_g4(m)
    g8_manifest(build(load_cfg("nerfsyn/chair.yml")))
    model_case("lego1k", load_cfg("nerfsyn/lego.yml", **small))
    tiny = {"geoms": {"points": {"init_num": 1000, "select_k": 12}},
            "models": {"use_renderer": False, "attn": {"d_model": 64, "embed": {
                "k_L": [4, 4, 4], "q_L": [4], "v_L": [4, 4],
                "key": {"d_ff": 64, "d_ff_out": 64, "n_ff_layer": 3},
                "query": {"d_ff": 64, "d_ff_out": 64, "n_ff_layer": 3},
                "value": {"d_ff": 64, "d_ff_out": 3, "n_ff_layer": 4}}}}}
    model_case("tiny_norender", load_cfg("nerfsyn/chair.yml", **tiny), n_img=2, hw=8, ray_seed=4)
    g7_trajectory(cfg1)
    g9_dp(cfg1)
    g8_init(cfg1, "g8_init_chair1k.npz")
    g8_init(load_cfg("nerfsyn/lego.yml", **small), "g8_init_lego1k.npz")
