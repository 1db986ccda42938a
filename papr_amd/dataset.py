"""Training / evaluation data for the drivers (counterpart of the reference's dataset/ package).

Three sources with one interface:
  * BlenderScene  -- a nerf_synthetic-format directory (transforms_{split}.json + PNGs), read with
                     PIL (imageio is not available here); semantics of dataset/load_nerfsyn.py and
                     dataset/dataset.py:10-121 (white background compositing, coord_scale applied to
                     the whole c2w, pixel-centre rays);
  * TanksTemplesScene -- a Tanks&Temples directory (dataset/load_t2.py);
  * SyntheticRayData (papr_amd/data.py) -- the procedural scene, used when the directory is absent.
Rays are generated on the device (papr_amd.data.get_rays); images stay resident on the device.
"""
import json
import math
import os

import numpy as np
import torch

from .data import SyntheticRayData, get_rays


class _ImageScene:
    """Images + poses resident on the device; patches as the reference's extract_patches draws them
    (dataset/utils.py:99-118: start row, then start column, uniform over [0, H - h) x [0, W - w)).

    Which image and which crop, in one process: the REFERENCE's own streams -- the image order of a `DataLoader(dataset, batch_size,
    shuffle)` started afresh every epoch (train.py:205-206: the sampler's permutation comes from the global torch generator) and the two
    `np.random.randint` draws of the GLOBAL numpy stream per patch (dataset/utils.py:110-111) -- so that one seed gives the reference's
    batch sequence and leaves the numpy stream where the reference's `add_points` finds it (golden G15, tests/test_dynamics_golden.py).
    own_stream (data parallelism; train.py passes it with seed = args.seed + rank): image and crop from this object's OWN numpy
    RandomState, so that every rank samples different patches while the global numpy stream stays identical on all ranks."""

    def _finish(self, images, c2w, focal_x, focal_y, dcfg, device, seed, own_stream=False):
        self.images = torch.from_numpy(np.ascontiguousarray(images)).float().to(device)
        s = dcfg["coord_scale"]
        c2w = torch.from_numpy(np.ascontiguousarray(c2w)).float()
        if s != 1:                                   # the whole matrix is scaled: rotation and translation (dataset.py:19-26)
            c2w = torch.matmul(torch.diag(torch.tensor([s, s, s, 1.0])), c2w)
        self.c2w = c2w.to(device)
        self.H, self.W = self.images.shape[1:3]
        self.focal_x, self.focal_y = float(focal_x), float(focal_y)
        self.focal = self.focal_x
        self.ph, self.pw = dcfg["patches"]["height"], dcfg["patches"]["width"]
        self.device = device
        # a quirk of the reference kept for the stream's sake: `args.patches.max_patches = 1` in RINDataset.__getitem__ (dataset/dataset.py:87-88) writes to a
        # temporary DictAsMember copy, so extract_patches still draws `patches.max_patches` (default.yml: 10) crops per item and the FIRST one is used
        self.spare_crops = 0 if own_stream else max(int(dcfg["patches"].get("max_patches", 1)) - 1, 0)
        self.rng = np.random.RandomState(seed) if own_stream else np.random
        self.own_stream = own_stream
        self.shuffle = bool(dcfg.get("shuffle", True))
        self._loader = self._epoch = None

    def next_indices(self, batch_size):
        """Image indices of the next batch, as the reference's train loop meets them."""
        if self.own_stream:
            self.last_indices = [int(self.rng.randint(0, len(self))) for _ in range(batch_size)]
            return self.last_indices
        if self._loader is None or self._loader.batch_size != batch_size:
            from torch.utils.data import DataLoader
            self._loader = DataLoader(range(len(self)), batch_size=batch_size, shuffle=self.shuffle, collate_fn=list)
            self._epoch = None
        while True:
            if self._epoch is None:
                self._epoch = iter(self._loader)
            try:
                self.last_indices = [int(i) for i in next(self._epoch)]
                return self.last_indices
            except StopIteration:
                self._epoch = None

    def __len__(self):
        return self.c2w.shape[0]

    def get_c2w(self, i):
        return self.c2w[i]

    def patch(self, img_idx=None):
        if img_idx is None:
            img_idx = self.next_indices(1)[0]
        h0 = int(self.rng.randint(0, self.H - self.ph))
        w0 = int(self.rng.randint(0, self.W - self.pw))
        for _ in range(self.spare_crops):
            self.rng.randint(0, self.H - self.ph), self.rng.randint(0, self.W - self.pw)
        c2w = self.c2w[img_idx:img_idx + 1]
        rayo, rayd = get_rays(self.H, self.W, self.focal_x, self.focal_y, c2w, h0, w0, self.ph, self.pw)
        tgt = self.images[img_idx:img_idx + 1, h0:h0 + self.ph, w0:w0 + self.pw]
        return tgt, rayd, rayo, c2w

    def full_view(self, img_idx, H=None, W=None):
        c2w = self.c2w[img_idx:img_idx + 1]
        rayo, rayd = get_rays(self.H, self.W, self.focal_x, self.focal_y, c2w)
        return self.images[img_idx:img_idx + 1], rayd, rayo, c2w


class BlenderScene(_ImageScene):
    """nerf_synthetic directory: transforms_{split}.json + PNGs (reference dataset/load_nerfsyn.py:8-42,
    dataset/utils.py:133-147: white-background compositing of RGBA)."""

    def __init__(self, dcfg, mode="train", device="cuda", seed=0, own_stream=False):
        from PIL import Image
        base = dcfg["path"]
        with open(os.path.join(base, "transforms_%s.json" % mode)) as f:
            meta = json.load(f)
        imgs, poses = [], []
        for fr in meta["frames"]:
            im = Image.open(os.path.join(base, fr["file_path"] + ".png"))
            if dcfg["factor"] > 1:
                im = im.resize((im.width // dcfg["factor"], im.height // dcfg["factor"]))
            a = (np.array(im) / 255.).astype(np.float32)
            if dcfg["white_bg"] and a.shape[-1] == 4:
                a = a[..., :3] * a[..., 3:] + (1.0 - a[..., 3:])
            else:
                a = a[..., :3]
            imgs.append(a)
            poses.append(np.array(fr["transform_matrix"], dtype=np.float32))
        W = imgs[0].shape[1]
        focal = .5 * W / np.tan(.5 * float(meta["camera_angle_x"]))
        self._finish(np.stack(imgs), np.stack(poses), focal, focal, dcfg, device, seed, own_stream)


_BLENDER2OPENCV = np.array([[1, 0, 0, 0], [0, -1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]])


def _t2_intrinsic(path):
    """3x3 intrinsics: a matrix file, or one line `f cx cy _` (reference dataset/load_t2.py:10-27)."""
    try:
        return np.loadtxt(path).astype(np.float32)[:3, :3]
    except ValueError:
        pass
    with open(path) as f:
        foc, cx, cy, _ = map(float, f.readline().split())
    return np.array([[foc, 0., cx], [0., foc, cy], [0., 0., 1.]])


class TanksTemplesScene(_ImageScene):
    """Tanks&Temples directory as the reference reads it (dataset/load_t2.py:29-86, dataset/utils.py:149-161):
    rgb/0_*.png train and rgb/1_*.png test views ordered by their trailing number, pose/<name>.txt OpenCV camera-to-world
    (flipped to the Blender convention), intrinsics.txt; with factor != 1 the images are resized to
    (2176 // factor) x (1280 // factor) and the focal lengths follow; without white_bg pure-white pixels become black."""

    def __init__(self, dcfg, mode="train", device="cuda", seed=0, tgtH=1280, tgtW=2176, own_stream=False):
        from PIL import Image
        base = dcfg["path"]
        colordir, posedir = os.path.join(base, "rgb"), os.path.join(base, "pose")
        lead = {"train": "0", "test": "1"}.get(mode)
        if lead is None:
            raise ValueError("Unknown split: {}".format(mode))
        names = [f for f in os.listdir(colordir) if os.path.isfile(os.path.join(colordir, f)) and f.startswith(lead)]
        names = sorted(names, key=lambda x: int(x.split(".")[0].split("_")[-1]))
        K = _t2_intrinsic(os.path.join(base, "intrinsics.txt"))
        fx, fy = K[0][0], K[1][1]
        imgs, poses = [], []
        for name in names:
            im = Image.open(os.path.join(colordir, name))
            W0, H0 = im.width, im.height
            if dcfg["factor"] != 1:
                im = im.resize((tgtW // dcfg["factor"], tgtH // dcfg["factor"]))
            imgs.append((np.array(im) / 255.).astype(np.float32))
            pose = np.loadtxt(os.path.join(posedir, name.replace(".png", ".txt"))).astype(np.float32)
            poses.append(pose @ _BLENDER2OPENCV)
        images = np.stack(imgs, 0)
        realH, realW = images.shape[1:3]
        fx, fy = fx * (realW / W0), fy * (realH / H0)
        if dcfg["white_bg"] and images.shape[-1] == 4:
            images = images[..., :3] * images[..., -1:] + (1. - images[..., -1:])
        elif not dcfg["white_bg"]:
            images = images[..., :3]
            images[images.sum(-1) == 3.0] = 0.
        self._finish(images, np.stack(poses, 0), fx, fy, dcfg, device, seed, own_stream)


class _Procedural(SyntheticRayData):
    def __len__(self):
        return self.c2w.shape[0]


def get_dataset(dcfg, mode="train", device="cuda", seed=0, views=None, own_stream=False):
    meta = os.path.join(dcfg["path"], "transforms_%s.json" % mode)
    if dcfg["type"] == "synthetic" and os.path.exists(meta):
        return BlenderScene(dcfg, mode, device, seed=seed, own_stream=own_stream)
    if dcfg["type"] == "t2" and os.path.isdir(os.path.join(dcfg["path"], "rgb")):
        return TanksTemplesScene(dcfg, mode, device, seed=seed, own_stream=own_stream)
    if dcfg["type"] not in ("synthetic", "t2"):
        raise ValueError("Unknown dataset type: {}".format(dcfg["type"]))
    print("[papr_amd] %s not found -> procedural scene (papr_amd/data.py)" % dcfg["path"])
    n = views or (100 if mode == "train" else 200)
    return _Procedural(dcfg, n_views=n, seed=seed + (0 if mode == "train" else 1000), device=device)


def sample_batch(dataset, batch_size):
    """`batch_size` patches stacked the way the reference's DataLoader collates them:
    tgt (N,h,w,3), rayd (N,h,w,3), rayo (N,3), c2w (N,4,4)."""
    if hasattr(dataset, "next_indices"):        # (the last batch of an epoch may be short, as the reference's DataLoader leaves it)
        parts = [dataset.patch(i) for i in dataset.next_indices(batch_size)]
    else:
        parts = [dataset.patch() for _ in range(batch_size)]
    return tuple(torch.cat([p[i] for p in parts], dim=0) for i in range(4))
