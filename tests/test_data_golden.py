"""Data path against the reference: ray generation, patch cropping, the Tanks&Temples reader.

Fixtures: tests/golden/g10_rays.npz (reference get_rays / extract_patches, dataset/utils.py:81-118) and g11_t2.npz
(reference RINDataset over a generated T&T-format scene, dataset/load_t2.py + dataset/dataset.py:10-47), both written by
tests/golden/make_golden.py --round2.  The CPU tests compare bit for bit; the device test allows 1 ulp of a unit vector."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import golden
from formula import write_t2_fixture

T = torch.from_numpy


def _samples(rd):
    N, H, W, _ = rd.shape
    return dict(rows=rd[:, [0, 1, H // 2, H - 1]], cols=rd[:, :, [0, 1, W // 2, W - 1]], block=rd[:, H // 2 - 16:H // 2 + 16, W // 3:W // 3 + 32])


def _check_rays(device, tol):
    from papr_amd.data import get_rays
    g = golden("g10_rays.npz")
    for tag in ("blender", "t2"):
        H, W, fx, fy, _ = g[tag + "/cam"]
        H, W = int(H), int(W)
        c2w = T(g[tag + "/c2w"]).to(device)
        ro, rd = get_rays(H, W, fx, fy, c2w)
        assert rd.shape == (2, H, W, 3) and torch.equal(ro.cpu(), T(g[tag + "/rays_o"]))
        for key, got in _samples(rd).items():
            ref = g[tag + "/" + key]
            err = np.abs(got.cpu().numpy() - ref).max()
            assert err <= tol, (tag, key, err)
        sums = np.stack([[rd[n].double().sum().item(), rd[n].double().abs().sum().item()] for n in range(2)])
        np.testing.assert_allclose(sums, g[tag + "/sums"][:, :2], rtol=1e-9 if tol == 0 else 1e-6)
        # the crop of extract_patches == a crop generated directly (what the training loop does on the device)
        ph = 160 if tag == "blender" else 180
        for i in range(2):
            for j in range(2):
                h0, w0 = (int(v) for v in g[tag + "/patch_hw"][i, j])
                _, crop = get_rays(H, W, fx, fy, c2w[i:i + 1], h0, w0, ph, ph)
                assert torch.equal(crop[0], rd[i, h0:h0 + ph, w0:w0 + ph])
                err = np.abs(crop[0, :4, :4].cpu().numpy() - g[tag + "/patch_rayd_corner"][i, j]).max()
                assert err <= tol, (tag, i, j, err)


def test_get_rays_reproduces_reference_bit_for_bit_on_cpu():
    _check_rays("cpu", 0.0)


@pytest.mark.gpu
def test_get_rays_on_device_within_one_ulp_of_reference():
    _check_rays("cuda", 1.2e-7)


def test_patch_offsets_follow_the_reference_numpy_draws(tmp_path):
    """extract_patches draws (start row, start column) per patch from numpy.  own_stream (data parallelism): a scene with seed 7 draws the crops the
    reference draws under np.random.seed(7) and never touches the global stream (add_points owns that one); one process: the crops come from
    the GLOBAL stream, like the reference's."""
    from papr_amd.dataset import _ImageScene
    g = golden("g10_rays.npz")
    for tag, ph in (("blender", 160), ("t2", 180)):
        H, W, fx, fy, scale = g[tag + "/cam"]
        H, W = int(H), int(W)
        ids = np.tile(np.arange(H * W, dtype=np.float32).reshape(1, H, W, 1), (2, 1, 1, 3))      # pixel id as colour: reveals the crop
        for own in (True, False):
            sc = _ImageScene()
            sc._finish(ids, np.tile(np.eye(4, dtype=np.float32), (2, 1, 1)), fx, fy,
                       {"coord_scale": 1.0, "patches": {"height": ph, "width": ph}}, "cpu", seed=7, own_stream=own)
            np.random.seed(123 if own else 7)
            state = np.random.get_state()[1].copy()
            got = []
            for i in range(2):
                for j in range(2):
                    tgt = sc.patch(i)[0]
                    first = int(tgt[0, 0, 0, 0])
                    got.append([first // W, first % W])
            assert np.array_equal(np.array(got).reshape(2, 2, 2), g[tag + "/patch_hw"])
            assert np.array_equal(np.random.get_state()[1], state) == own


def test_tanks_and_temples_reader_matches_reference_dataset(tmp_path):
    from papr_amd.dataset import get_dataset
    g = golden("g11_t2.npz")
    base = str(tmp_path / "scene")
    write_t2_fixture(base)
    dcfg = {"type": "t2", "path": base, "factor": 1, "white_bg": False, "coord_scale": 30.0, "patches": {"height": 8, "width": 8}}
    for mode in ("train", "test"):
        ds = get_dataset(dcfg, mode, "cpu", seed=0)
        assert type(ds).__name__ == "TanksTemplesScene"
        H, W, fx, fy = g[mode + "/hwf"]
        assert (ds.H, ds.W) == (int(H), int(W)) and ds.focal_x == fx and ds.focal_y == fy
        assert torch.equal(ds.c2w, T(g[mode + "/c2w"]))
        assert np.array_equal(ds.images[0, :6, :8].numpy(), g[mode + "/image0_head"])
        sums = np.stack([[ds.images[i].double().sum().item(), ds.images[i].double().abs().sum().item()] for i in range(len(ds))])
        np.testing.assert_allclose(sums, g[mode + "/images_sums"][:, :2], rtol=1e-12)
        assert torch.all(ds.images[:, :4, :5] == 0)                # white -> black without white_bg (dataset/utils.py:157-160)
        img, rayd, rayo, c2w = ds.full_view(0)
        assert torch.equal(rayd[0], T(g[mode + "/rayd0"])) and torch.equal(rayo[0], T(g[mode + "/rayo"])[0])
        tgt, prd, pro, pc = ds.patch()
        assert tgt.shape == (1, 8, 8, 3) and prd.shape == (1, 8, 8, 3)


def _write_blender_fixture(base, n=5, H=24, W=24):
    from PIL import Image
    from papr_amd.data import make_cameras
    os.makedirs(os.path.join(base, "train"), exist_ok=True)
    cams = make_cameras(n, seed=2, coord_scale=1.0).numpy()
    rs = np.random.RandomState(0)
    frames = []
    for i in range(n):
        Image.fromarray(rs.randint(0, 256, (H, W, 4)).astype(np.uint8)).save(os.path.join(base, "train", "r_%d.png" % i))
        frames.append({"file_path": "./train/r_%d" % i, "transform_matrix": cams[i].tolist()})
    json.dump({"camera_angle_x": 0.6911112070083618, "frames": frames}, open(os.path.join(base, "transforms_train.json"), "w"))


def test_ranks_draw_different_patches_from_a_blender_directory(tmp_path):
    """Data parallelism: train.py passes seed + rank; each rank must sample its own images / crops (otherwise N GPUs average
    N copies of one gradient), while the global numpy stream stays rank-independent for add_points."""
    from papr_amd.dataset import get_dataset
    base = str(tmp_path / "lego")
    _write_blender_fixture(base)
    dcfg = {"type": "synthetic", "path": base, "factor": 1, "white_bg": True, "coord_scale": 10.0, "patches": {"height": 8, "width": 8}}
    np.random.seed(11)
    state = np.random.get_state()[1].copy()
    r0, r1, r0_again = (get_dataset(dcfg, "train", "cpu", seed=s, own_stream=True) for s in (5, 6, 5))
    assert type(r0).__name__ == "BlenderScene" and r0.images.shape == (5, 24, 24, 3)
    a = [r0.patch() for _ in range(6)]
    b = [r1.patch() for _ in range(6)]
    c = [r0_again.patch() for _ in range(6)]
    assert any(not torch.equal(x[1], y[1]) or not torch.equal(x[0], y[0]) for x, y in zip(a, b)), "ranks sample identical patches"
    assert all(torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) for x, y in zip(a, c)), "same seed must give the same patches"
    assert np.array_equal(np.random.get_state()[1], state), "the global numpy stream (add_points) must not be consumed by the sampler"
    # white-background compositing of RGBA (dataset/utils.py:141-143) and the c2w scaling (dataset/dataset.py:19-26)
    assert float(r0.images.min()) >= 0 and float(r0.images.max()) <= 1
    assert abs(float(r0.c2w[0, :3, 3].norm()) - 10.0 * 4.031128874) < 1e-3
