"""Image metrics of the evaluation driver (reference test.py:107-110).

PSNR as in test.py:107.  SSIM restates what the reference calls --
`skimage.metrics.structural_similarity(gt, img, win_size=11, channel_axis=2, data_range=1.0)` (test.py:23-24) with that
function's defaults: uniform 11 x 11 window (scipy.ndimage.uniform_filter, reflecting borders), sample covariance,
K1 = 0.01, K2 = 0.03, float64, the (win-1)/2 border of the SSIM map dropped before averaging, mean over channels
(Wang et al., "Image quality assessment: from error visibility to structural similarity", 2004).  scikit-image is not
available in this environment, so the restatement is not pinned against it; tests check its defining properties.
LPIPS: papr_amd/lpips.py (TestLPIPS), computed when the pretrained weights are at hand.  depth_map: test.py:113-119.
"""
import numpy as np
from scipy.ndimage import uniform_filter


def psnr(rgb, img):
    return float(-10.0 * np.log(np.mean((np.asarray(rgb, np.float64) - np.asarray(img, np.float64)) ** 2)) / np.log(10.0))


def ssim(gt, img, win_size=11, data_range=1.0, k1=0.01, k2=0.03):
    """gt, img: (H, W, C) arrays in [0, data_range]; returns the mean SSIM over pixels and channels."""
    gt = np.asarray(gt, np.float64)
    img = np.asarray(img, np.float64)
    assert gt.shape == img.shape and gt.ndim == 3 and min(gt.shape[:2]) >= win_size and win_size % 2 == 1
    npix = win_size ** 2
    cov_norm = npix / (npix - 1.0)                      # sample covariance
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    pad = (win_size - 1) // 2
    vals = []
    for ch in range(gt.shape[2]):
        x, y = gt[..., ch], img[..., ch]
        ux, uy = uniform_filter(x, size=win_size), uniform_filter(y, size=win_size)
        uxx, uyy, uxy = uniform_filter(x * x, size=win_size), uniform_filter(y * y, size=win_size), uniform_filter(x * y, size=win_size)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
        vals.append(s[pad:-pad, pad:-pad].mean())
    return float(np.mean(vals))


def depth_map(selected_points, attn, rays_o):
    """Expected depth of a rendered view as the reference's test_step forms it (test.py:113-119): distance of every selected point
    from the camera plane through the origin of the rays (normal -o, offset |o|^2), weighted with the attention over the k
    points (the background token counts with distance 0).  selected_points (1,H,W,k,3), attn (1,H,W,k+1,1), rays_o (1,3) torch
    tensors -> (H,W) float32 numpy, scene units (the PNG scales by 65536 / 10 / coord_scale, test.py:126-128)."""
    import torch
    od = -rays_o
    D = torch.sum(od * rays_o)
    dists = torch.abs(torch.sum(selected_points.to(od.device) * od, -1) - D) / torch.norm(od)
    dists = torch.cat([dists, torch.zeros_like(dists[..., :1])], dim=-1)
    return torch.sum(attn.squeeze(-1).to(od.device) * dists, dim=-1).detach().cpu().squeeze().numpy().astype(np.float32)
