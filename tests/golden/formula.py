"""Deterministic parameter / input fill shared by the golden generator and the tests.

The golden fixtures store *outputs* of the reference for models whose parameters are
produced by ``formula_fill`` (seeded torch CPU generators keyed by the parameter name),
so multi-million-parameter models need no stored weights.  torch's CPU mt19937 stream is
stable across machines for a given torch version (the GPU box runs the same image).
"""
import zlib

import torch


def _gen(name, salt=0):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) + 7919 * salt) % (2 ** 31))
    return g


def formula_tensor(name, shape, salt=0):
    g = _gen(name, salt)
    shape = tuple(shape)
    u = lambda lo, hi: torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo
    if name == "pc_feats":
        return torch.randn(shape, generator=g, dtype=torch.float32)
    if name == "points_influ_scores":
        return u(-0.2, 1.0)
    if name.endswith("a_2"):
        return u(0.8, 1.2)
    if name.endswith("b_2"):
        return u(-0.1, 0.1)
    if len(shape) > 1:
        rf = 1
        for s in shape[2:]:
            rf *= s
        fan_in, fan_out = shape[1] * rf, shape[0] * rf
        b = (6.0 / (fan_in + fan_out)) ** 0.5
        return u(-b, b)
    return u(-0.05, 0.05)


SKIP = ("select_k", "bkg_feats", "points")


def formula_fill(state_dict, salt=0):
    """In-place fill of every float tensor except points / bkg_feats / select_k."""
    with torch.no_grad():
        for name, t in state_dict.items():
            if name in SKIP or not t.is_floating_point():
                continue
            t.copy_(formula_tensor(name, t.shape, salt))
    return state_dict


def uniform_points(P, half_extent, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return (torch.rand((P, 3), generator=g, dtype=torch.float32) * 2 - 1) * half_extent


def synth_rays(n_img, H, W, seed=0, radius=40.0, spread=0.05):
    """Seeded camera rays looking at the origin from `radius` away (unit directions).

    Image 0 sits on +z; further images are rotated about the y axis.  Returns
    rays_o (n,3), rays_d (n,H,W,3), c2w (n,4,4) (c2w is accepted but unused by the path).
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    rays_o, rays_d = [], []
    for i in range(n_img):
        ang = 0.9 * i
        ca, sa = torch.cos(torch.tensor(ang)), torch.sin(torch.tensor(ang))
        rot = torch.tensor([[ca, 0.0, sa], [0.0, 1.0, 0.0], [-sa, 0.0, ca]], dtype=torch.float32)
        d = torch.randn((H, W, 3), generator=g, dtype=torch.float32) * spread + torch.tensor([0.0, 0.0, -1.0])
        d = d / d.norm(dim=-1, keepdim=True)
        rays_d.append(d @ rot.T)
        rays_o.append(rot @ torch.tensor([0.0, 0.0, radius]))
    c2w = torch.eye(4).repeat(n_img, 1, 1)
    return torch.stack(rays_o), torch.stack(rays_d), c2w


def write_t2_fixture(base, H=40, W=64, n_train=3, n_test=2, seed=3):
    """A Tanks&Temples-format scene (dataset/load_t2.py): rgb/{0,1}_*.png, pose/*.txt (OpenCV c2w), intrinsics.txt."""
    import os
    import numpy as np
    from PIL import Image
    from papr_amd.data import make_cameras
    os.makedirs(os.path.join(base, "rgb"), exist_ok=True)
    os.makedirs(os.path.join(base, "pose"), exist_ok=True)
    rs = np.random.RandomState(seed)
    K = np.array([[55.5, 0, W / 2.0, 0], [0, 56.25, H / 2.0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    np.savetxt(os.path.join(base, "intrinsics.txt"), K)
    cams = make_cameras(n_train + n_test, seed=seed, coord_scale=1.0).numpy().astype(np.float64)
    flip = np.diag([1.0, -1.0, -1.0, 1.0])
    for i in range(n_train + n_test):
        name = "%d_%04d_%08d" % (0 if i < n_train else 1, i, i)
        img = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
        img[:4, :5] = 255                                                   # pure white pixels: zeroed when white_bg is false
        Image.fromarray(img).save(os.path.join(base, "rgb", name + ".png"))
        np.savetxt(os.path.join(base, "pose", name + ".txt"), cams[i] @ flip)   # stored in OpenCV convention


def write_blender_fixture(base, res=40, n_train=12, n_test=3, seed=5):
    """A nerf_synthetic-format scene (dataset/load_nerfsyn.py): transforms_{train,test}.json + RGBA PNGs of two shaded spheres on a transparent
    background, ray-traced in numpy from orbit cameras with the dataset's field of view.  Deterministic: the golden generator and the tests
    write the same bytes."""
    import json
    import os
    import numpy as np
    from PIL import Image
    from papr_amd.data import CAMERA_ANGLE_X, make_cameras
    cams = make_cameras(n_train + n_test, seed=seed, coord_scale=1.0).numpy().astype(np.float64)
    focal = 0.5 * res / np.tan(0.5 * CAMERA_ANGLE_X)
    px = (np.arange(res) + 0.5 - res / 2.0) / focal
    x, y = np.meshgrid(px, -px)
    dirs = np.stack([x, y, -np.ones_like(x)], -1)
    spheres = [(np.array([0.0, 0.0, 0.0]), 0.9, np.array([0.85, 0.35, 0.2])), (np.array([0.7, -0.5, 0.6]), 0.45, np.array([0.2, 0.5, 0.9]))]
    light = np.array([0.4, 0.3, 0.85]) / np.linalg.norm([0.4, 0.3, 0.85])
    for split, lo, hi in (("train", 0, n_train), ("test", n_train, n_train + n_test)):
        os.makedirs(os.path.join(base, split), exist_ok=True)
        frames = []
        for i in range(lo, hi):
            c2w = cams[i]
            d = dirs @ c2w[:3, :3].T
            d = d / np.linalg.norm(d, axis=-1, keepdims=True)
            o = c2w[:3, 3]
            rgba = np.zeros((res, res, 4))
            depth = np.full((res, res), np.inf)
            for c, r, col in spheres:
                oc = o - c
                b = (d * oc).sum(-1)
                disc = b * b - (oc @ oc - r * r)
                t = -b - np.sqrt(np.maximum(disc, 0.0))
                hit = (disc > 0) & (t > 0) & (t < depth)
                n = (o + d * t[..., None] - c) / r
                shade = 0.25 + 0.75 * np.maximum((n * light).sum(-1), 0.0)
                rgba[hit, :3] = (col * shade[..., None])[hit]
                rgba[hit, 3] = 1.0
                depth = np.where(hit, t, depth)
            name = "r_%d" % (i - lo)
            Image.fromarray(np.round(rgba * 255).astype(np.uint8), "RGBA").save(os.path.join(base, split, name + ".png"))
            frames.append({"file_path": "./%s/%s" % (split, name), "transform_matrix": [[float(v) for v in row] for row in c2w]})
        with open(os.path.join(base, "transforms_%s.json" % split), "w") as f:
            json.dump({"camera_angle_x": CAMERA_ANGLE_X, "frames": frames}, f)
