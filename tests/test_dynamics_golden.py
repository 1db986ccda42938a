"""Training DYNAMICS against the reference's own loop (golden G15, tests/golden/g15_dynamics.npz: /root/reference/train.py:182-300
train_and_eval -- DataLoader, train_step, prune / add schedule, init_optimizers(step), eval_step -- run on CPU by tests/golden/make_golden.py
--g15 for 360 steps of the tiny nerf_synthetic-format directory formula.write_blender_fixture writes).

CPU half: the batch SEQUENCE.  One seed gives the reference's image order (its DataLoader's shuffles) and crops (the global numpy stream), and
leaves both streams where the reference's next draws find them.
GPU half (`-m gpu`): `train.py` on the same directory and seed -- the same images and crops step by step, the same events at the same steps,
point counts / loss curve / eval PSNR within bands stated against how far the reference drifts from ITSELF between two CPU thread counts."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden
from formula import write_blender_fixture


def _overrides(scene, save_dir):
    g = golden("g15_dynamics.npz")
    over = json.loads(str(g["cfg_json"]))
    over["dataset"]["path"] = scene
    over["eval"]["dataset"] = {"path": scene}
    over.update(save_dir=save_dir, index="g15", use_amp=False)
    over["training"]["losses"] = {"mse": 1.0, "lpips": 0.0, "lpips_alex": 0.0}
    return g, over


def _flat(d, prefix=""):
    out = []
    for k, v in d.items():
        if isinstance(v, dict):
            out += _flat(v, prefix + k + ".")
        else:
            out.append("%s%s=%s" % (prefix, k, json.dumps(v) if not isinstance(v, str) else v))
    return out


def test_one_seed_gives_the_reference_batch_sequence(tmp_path):
    """train.py's sampling (papr_amd.dataset: next_indices + patch) after the seeded model construction: image index and crop of all 360 steps."""
    from papr_amd import get_model, load_config
    from papr_amd.dataset import get_dataset, sample_batch
    scene = str(tmp_path / "scene") + "/"
    write_blender_fixture(scene)
    g, over = _overrides(scene, str(tmp_path / "exp"))
    cfg = load_config("nerfsyn/chair.yml", overrides=over)
    import random
    torch.manual_seed(cfg["seed"]); np.random.seed(cfg["seed"]); random.seed(cfg["seed"])
    get_model(cfg, device="cpu")                                  # consumes the torch / numpy streams exactly as the reference's constructor does
    ds = get_dataset(cfg["dataset"], "train", "cpu", seed=cfg["seed"])
    assert type(ds).__name__ == "BlenderScene" and len(ds) == 12
    imgs, sums = [], []
    for step in range(360):
        tgt, rayd, rayo, c2w = sample_batch(ds, cfg["dataset"]["batch_size"])
        if step == 240:                                           # the reference's growth step draws from the global numpy stream here (80 x 3 uniforms),
            np.random.uniform(0, 1, (80, 3))                      # with the step's batch already in hand (`for batch in trainloader`)
        imgs.append(ds.last_indices[0])
        sums.append(float(tgt.double().sum()))
    assert np.array_equal(np.array(imgs), g["a/img"])
    # (the reference run's prune at step 240 left 464 points > add_num: sites by ranking, no site draw -- the 240 uniforms above are all it consumed)
    np.testing.assert_allclose(np.array(sums), g["a/tgt_sum"], rtol=1e-12)


@pytest.mark.gpu
def test_train_py_follows_the_reference_training_dynamics(tmp_path):
    scene = str(tmp_path / "scene") + "/"
    write_blender_fixture(scene)
    g, over = _overrides(scene, str(tmp_path / "exp"))
    log = str(tmp_path / "steps.npz")
    cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--opt", os.path.join(ROOT, "configs", "nerfsyn", "chair.yml"), "--log-steps", log, "--set"] + _flat(over)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    t = np.load(log)
    a_loss, b_loss = g["a/loss"], g["b/loss"]
    # 1. the same data: image order and crops, step by step, across the prune / add events
    assert np.array_equal(t["img"], g["a/img"])
    np.testing.assert_allclose(t["tgt_sum"], g["a/tgt_sum"], rtol=1e-6)
    # 2. before anything can drift: the first steps are the reference's losses
    assert np.abs(t["loss"][:20] - a_loss[:20]).max() <= 2e-5, np.abs(t["loss"][:20] - a_loss[:20]).max()
    # 3. events: the same kinds at the same steps; counts within the band.  Yardstick: the reference against itself (8 vs 3 CPU threads) differs
    #    by `ref_ev` points at an event
    ev, ev_a, ev_b = t["events"], g["a/events"], g["b/events"]
    assert np.array_equal(ev[:, :2], ev_a[:, :2]), (ev, ev_a)
    ref_ev = int(np.abs(ev_a[:, 2:] - ev_b[:, 2:]).max())
    got_ev = int(np.abs(ev[:, 2:] - ev_a[:, 2:]).max())
    print("event counts: build vs reference differ by at most %d points, reference vs itself by %d" % (got_ev, ref_ev), ev.tolist())
    assert np.array_equal(ev[:2], ev_a[:2]), "the first two prune events (before the trajectories can drift apart) must agree exactly"
    assert got_ev <= max(4 * ref_ev, 8)
    # 4. loss curve in 20-step means, against the reference's own drift
    m = lambda x: x.reshape(-1, 20).mean(1)
    ref_d = np.abs(m(a_loss) - m(b_loss))
    got_d = np.abs(m(t["loss"]) - m(a_loss))
    print("20-step loss means: build vs reference max %.2e, reference vs itself max %.2e" % (got_d.max(), ref_d.max()))
    assert got_d[:6].max() <= 1e-4                                # the first 120 steps (no event yet)
    assert got_d.max() <= max(5 * ref_d.max(), 3e-3)
    # 5. evaluation PSNR at steps 100 / 200 / 300 / 360 (test.py:107 formula on the chunked full-image render)
    ps, ps_a, ps_b = t["eval_psnrs"], g["a/eval_psnrs"], g["b/eval_psnrs"]
    print("eval PSNR build", ps.tolist(), "reference", ps_a.tolist(), "reference, other thread count", ps_b.tolist())
    assert len(ps) >= len(ps_a)
    assert abs(ps[0] - ps_a[0]) <= 0.01                           # step 100: before any event
    assert np.abs(ps[:len(ps_a)] - ps_a).max() <= max(5 * np.abs(ps_a - ps_b).max(), 0.15)
    assert abs(float(t["attn_lr"]) - float(g["a/attn_lr"])) <= 1e-12 and abs(float(t["pts_lr"]) - float(g["a/pts_lr"])) <= 1e-12
