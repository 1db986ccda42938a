"""Procedural stand-in for the nerf_synthetic scenes, generated directly in device memory.

No dataset can be downloaded in this environment, so benchmarks and the PSNR check train on an
analytically rendered object seen from Blender-style orbit cameras with the dataset's own
intrinsics (800x800, camera_angle_x = 0.6911112, radius 4.031, coord_scale 10; reference
dataset/load_nerfsyn.py:37-40, dataset/dataset.py:18-26).  Ray generation follows the reference's
convention (dataset/utils.py:81-96): pixel-centre directions (x, -y, -1)/focal in camera space,
rotated by c2w, normalised; origin = c2w translation.  It runs on the GPU, so the loader the
reference runs on the CPU every step is off the critical path.
"""
import math

import torch

CAMERA_ANGLE_X = 0.6911112070083618
ORBIT_RADIUS = 4.031128874


def orbit_pose(azimuth, elevation, radius):
    """Blender-convention camera-to-world (camera looks along -z, +y up), looking at the origin."""
    ca, sa, ce, se = math.cos(azimuth), math.sin(azimuth), math.cos(elevation), math.sin(elevation)
    eye = torch.tensor([radius * ce * ca, radius * ce * sa, radius * se])
    fwd = -eye / eye.norm()
    up = torch.tensor([0.0, 0.0, 1.0])
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm()
    true_up = torch.linalg.cross(right, fwd)
    c2w = torch.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, true_up, -fwd, eye
    return c2w


def make_cameras(n, seed, coord_scale=10.0, upper_only=True):
    g = torch.Generator().manual_seed(seed)
    poses = []
    for _ in range(n):
        az = float(torch.rand((), generator=g)) * 2 * math.pi
        el = float(torch.rand((), generator=g)) * (math.pi / 2 * 0.9) if upper_only else \
            (float(torch.rand((), generator=g)) - 0.5) * math.pi * 0.9
        poses.append(orbit_pose(az, el, ORBIT_RADIUS))
    c2w = torch.stack(poses)
    scale = torch.diag(torch.tensor([coord_scale, coord_scale, coord_scale, 1.0]))
    return scale @ c2w          # the reference scales rotation and translation alike (dataset.py:20-26)


_AXIS_CACHE = {}


def _pixel_axes(H, W, focal_x, focal_y):
    """Camera-space x (W,) and y (H,) of the pixel centres, computed exactly as the reference does
    (dataset/utils.py:83-91: linspace over the image plane, minus half the extent, plus half a pixel) -- on the CPU, so
    that the values do not depend on the device's linspace kernel; (H + W) floats, cached per camera."""
    key = (int(H), int(W), float(focal_x), float(focal_y))
    if key not in _AXIS_CACHE:
        width = torch.linspace(0, W / focal_x, steps=int(W) + 1, dtype=torch.float32)
        height = torch.linspace(0, H / focal_y, steps=int(H) + 1, dtype=torch.float32)
        px, py = width[1] - width[0], height[1] - height[0]
        _AXIS_CACHE[key] = ((width - W / focal_x / 2 + px / 2)[:-1].contiguous(), (-(height - H / focal_y / 2 + py / 2))[:-1].contiguous())
    return _AXIS_CACHE[key]


def get_rays(H, W, focal_x, focal_y, c2w, h0=0, w0=0, h=None, w=None):
    """rays_o (N,3), rays_d (N,h,w,3) for the crop [h0:h0+h, w0:w0+w] of an HxW view; any device.

    Counterpart of the reference's get_rays (dataset/utils.py:81-96) fused with the crop of extract_patches
    (:99-118): directions (x, y, -1) through the pixel centres, rotated by c2w (cam_to_world with a zero homogeneous
    coordinate: sum_j dir_j * c2w[i, j] in the reference's order), normalised; origin = the translation column."""
    h = H if h is None else h
    w = W if w is None else w
    dev = c2w.device
    xs, ys = _pixel_axes(H, W, focal_x, focal_y)
    xs, ys = xs[w0:w0 + w].to(dev), ys[h0:h0 + h].to(dev)
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    r = c2w[:, :3, :3].reshape(-1, 1, 1, 3, 3)
    # ((x r_i0 + y r_i1) + (-1) r_i2) + 0 r_i3: the four-term sum of cam_to_world, written out so that every device adds in one order
    rays_d = (xx[None, :, :, None] * r[..., 0] + yy[None, :, :, None] * r[..., 1]) + (-1.0) * r[..., 2]
    rays_d = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
    return c2w[:, :3, 3].contiguous(), rays_d.contiguous()


class SphereScene:
    """A chair-sized cluster of shaded spheres on a white background (analytic ground truth)."""

    def __init__(self, coord_scale=10.0, seed=0, n_spheres=7, device="cpu"):
        g = torch.Generator().manual_seed(seed)
        self.centers = ((torch.rand(n_spheres, 3, generator=g) - 0.5) * 1.2 * coord_scale).to(device)
        self.radii = ((0.18 + 0.22 * torch.rand(n_spheres, generator=g)) * coord_scale).to(device)
        self.colors = (0.15 + 0.8 * torch.rand(n_spheres, 3, generator=g)).to(device)
        self.light = torch.nn.functional.normalize(torch.tensor([0.4, -0.3, 0.85]), dim=0).to(device)

    def render(self, rays_o, rays_d):
        """rays_o (N,3), rays_d (N,h,w,3) -> rgb (N,h,w,3) in [0,1]."""
        N = rays_d.shape[0]
        o = rays_o.reshape(N, 1, 1, 1, 3)
        d = rays_d.unsqueeze(-2)                                    # (N,h,w,1,3)
        oc = o - self.centers                                       # (N,1,1,S,3)
        b = (oc * d).sum(-1)
        c = (oc * oc).sum(-1) - self.radii ** 2
        disc = b * b - c
        t = -b - torch.sqrt(disc.clamp_min(0))
        t = torch.where((disc > 0) & (t > 0), t, torch.full_like(t, float("inf")))
        tmin, which = t.min(-1)
        hit = torch.isfinite(tmin)
        p = rays_o.reshape(N, 1, 1, 3) + rays_d * torch.where(hit, tmin, torch.zeros_like(tmin)).unsqueeze(-1)
        nrm = torch.nn.functional.normalize(p - self.centers[which], dim=-1)
        shade = 0.35 + 0.65 * (nrm * self.light).sum(-1).clamp_min(0)
        rgb = self.colors[which] * shade.unsqueeze(-1)
        return torch.where(hit.unsqueeze(-1), rgb, torch.ones_like(rgb))


class SyntheticRayData:
    """Patch sampler with the reference's item layout (img_idx, patch_idx, tgt, rayd, rayo)
    (dataset/dataset.py:97-100) whose rays and targets are produced on `device`."""

    def __init__(self, dcfg, n_views=100, seed=0, device="cuda", H=800, W=800):
        self.H, self.W = H // dcfg["factor"], W // dcfg["factor"]
        self.focal = 0.5 * self.W / math.tan(0.5 * CAMERA_ANGLE_X)
        self.c2w = make_cameras(n_views, seed, dcfg["coord_scale"]).to(device)
        self.scene = SphereScene(dcfg["coord_scale"], seed=1234, device=device)
        self.ph, self.pw = dcfg["patches"]["height"], dcfg["patches"]["width"]
        self.device = device
        self.gen = torch.Generator().manual_seed(seed + 17)

    def get_c2w(self, i):
        return self.c2w[i]

    def patch(self, img_idx=None):
        """One random training patch: tgt (1,h,w,3), rayd (1,h,w,3), rayo (1,3), c2w (1,4,4)."""
        if img_idx is None:
            img_idx = int(torch.randint(0, self.c2w.shape[0], (), generator=self.gen))
        h0 = int(torch.randint(0, self.H - self.ph, (), generator=self.gen))
        w0 = int(torch.randint(0, self.W - self.pw, (), generator=self.gen))
        c2w = self.c2w[img_idx:img_idx + 1]
        rayo, rayd = get_rays(self.H, self.W, self.focal, self.focal, c2w, h0, w0, self.ph, self.pw)
        return self.scene.render(rayo, rayd), rayd, rayo, c2w

    def full_view(self, img_idx, H=None, W=None):
        H, W = H or self.H, W or self.W
        f = 0.5 * W / math.tan(0.5 * CAMERA_ANGLE_X)
        c2w = self.c2w[img_idx:img_idx + 1]
        rayo, rayd = get_rays(H, W, f, f, c2w)
        return self.scene.render(rayo, rayd), rayd, rayo, c2w
