"""Training / evaluation data for the drivers (counterpart of the reference's dataset/ package).

Two sources with one interface:
  * BlenderScene  -- a nerf_synthetic-format directory (transforms_{split}.json + PNGs), read with
                     PIL (imageio is not available here); semantics of dataset/load_nerfsyn.py and
                     dataset/dataset.py:10-121 (white background compositing, coord_scale applied to
                     the whole c2w, pixel-centre rays);
  * SyntheticRayData (papr_amd/data.py) -- the procedural scene, used when the directory is absent.
Rays are generated on the device (papr_amd.data.get_rays); images stay resident on the device.
"""
import json
import math
import os

import numpy as np
import torch

from .data import SyntheticRayData, get_rays


class BlenderScene:
    def __init__(self, dcfg, mode="train", device="cuda"):
        from PIL import Image
        base = dcfg["path"]
        with open(os.path.join(base, "transforms_%s.json" % mode)) as f:
            meta = json.load(f)
        imgs, poses = [], []
        for fr in meta["frames"]:
            im = Image.open(os.path.join(base, fr["file_path"] + ".png"))
            if dcfg["factor"] > 1:
                im = im.resize((im.width // dcfg["factor"], im.height // dcfg["factor"]))
            a = np.asarray(im, dtype=np.float32) / 255.0
            if dcfg["white_bg"] and a.shape[-1] == 4:
                a = a[..., :3] * a[..., 3:] + (1.0 - a[..., 3:])
            else:
                a = a[..., :3]
            imgs.append(a)
            poses.append(np.array(fr["transform_matrix"], dtype=np.float32))
        self.images = torch.from_numpy(np.stack(imgs)).to(device)
        c2w = torch.from_numpy(np.stack(poses))
        s = dcfg["coord_scale"]
        self.c2w = (torch.diag(torch.tensor([s, s, s, 1.0])) @ c2w).to(device)
        self.H, self.W = self.images.shape[1:3]
        self.focal = 0.5 * self.W / math.tan(0.5 * float(meta["camera_angle_x"]))
        self.ph, self.pw = dcfg["patches"]["height"], dcfg["patches"]["width"]
        self.device = device

    def __len__(self):
        return self.c2w.shape[0]

    def get_c2w(self, i):
        return self.c2w[i]

    def patch(self, img_idx=None):
        if img_idx is None:
            img_idx = np.random.randint(0, len(self))
        h0 = np.random.randint(0, self.H - self.ph)      # same draws as the reference's extract_patches
        w0 = np.random.randint(0, self.W - self.pw)
        c2w = self.c2w[img_idx:img_idx + 1]
        rayo, rayd = get_rays(self.H, self.W, self.focal, self.focal, c2w, h0, w0, self.ph, self.pw)
        tgt = self.images[img_idx:img_idx + 1, h0:h0 + self.ph, w0:w0 + self.pw]
        return tgt, rayd, rayo, c2w

    def full_view(self, img_idx, H=None, W=None):
        c2w = self.c2w[img_idx:img_idx + 1]
        rayo, rayd = get_rays(self.H, self.W, self.focal, self.focal, c2w)
        return self.images[img_idx:img_idx + 1], rayd, rayo, c2w


class _Procedural(SyntheticRayData):
    def __len__(self):
        return self.c2w.shape[0]


def get_dataset(dcfg, mode="train", device="cuda", seed=0, views=None):
    meta = os.path.join(dcfg["path"], "transforms_%s.json" % mode)
    if dcfg["type"] == "synthetic" and os.path.exists(meta):
        return BlenderScene(dcfg, mode, device)
    if dcfg["type"] == "t2":
        raise NotImplementedError("papr_amd: the Tanks&Temples loader is not built; use a nerf_synthetic-format directory")
    print("[papr_amd] %s not found -> procedural scene (papr_amd/data.py)" % meta)
    n = views or (100 if mode == "train" else 200)
    return _Procedural(dcfg, n_views=n, seed=seed + (0 if mode == "train" else 1000), device=device)


def sample_batch(dataset, batch_size):
    """`batch_size` patches stacked the way the reference's DataLoader collates them:
    tgt (N,h,w,3), rayd (N,h,w,3), rayo (N,3), c2w (N,4,4)."""
    parts = [dataset.patch() for _ in range(batch_size)]
    return tuple(torch.cat([p[i] for p in parts], dim=0) for i in range(4))
