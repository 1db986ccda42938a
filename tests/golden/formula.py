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
