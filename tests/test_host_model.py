"""Host-side behaviour of papr_amd.PAPR that needs no GPU: construction, state-dict surface,
initialisation stream, optimizers/schedules, prune/add, checkpoints, config merge, C-ABI exports."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT, case_cfg, golden


def seed_all(s):
    import random
    torch.manual_seed(s)
    np.random.seed(s)
    random.seed(s)


def build(tag="chair1k", seed=1):
    from papr_amd import get_model
    seed_all(seed)
    return get_model(tag if isinstance(tag, dict) else case_cfg(tag), device="cpu")


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), t.norm().item(), t.abs().max().item()])


def test_state_dict_matches_reference_manifest():
    from papr_amd import get_model, load_config
    man = json.load(open(os.path.join(GOLDEN, "g8_manifest.json")))
    seed_all(1)
    m = get_model(load_config("nerfsyn/chair.yml", overrides={"training": {"losses": {"lpips": 0.0}}}), device="cpu")
    sd = m.state_dict()
    ref = {k: v for k, v in man.items() if not k.startswith("__")}
    assert list(sd.keys()) == list(dict(sorted(ref.items())).keys()) or set(sd.keys()) == set(ref.keys())
    for k, (shape, dtype) in ref.items():
        assert list(sd[k].shape) == shape, k
        assert str(sd[k].dtype).replace("torch.", "") == dtype, k
    assert sorted(m.optimizers.keys()) == man["__optimizers__"]
    assert sum(p.numel() for p in m.proximity_attn.parameters()) == man["__counts__"]["attn"]
    assert sum(p.numel() for p in m.renderer.parameters()) == man["__counts__"]["renderer"]


@pytest.mark.parametrize("tag", ["chair1k", "lego1k"])
def test_initialisation_reproduces_reference_rng_stream(tag):
    """Same seed -> the same initial parameters as the reference (same construction order)."""
    g = golden("g8_init_%s.npz" % tag)
    m = build(tag)
    sd = m.state_dict()
    np.testing.assert_array_equal(sd["points"].numpy(), g["points"])
    np.testing.assert_array_equal(sd["pc_feats"][:8].numpy(), g["pc_feats_head"])
    np.testing.assert_array_equal(sd["proximity_attn.attention_layer.w_k.weight"][:4].numpy(), g["wk_head"])
    np.testing.assert_array_equal(sd["renderer.inc.double_conv.0.bias"].numpy(), g["inc_bias"])
    for name, ref in zip(g["names"], g["stats"]):
        np.testing.assert_allclose(stats(sd[str(name)]), ref, rtol=1e-12, atol=0, err_msg=str(name))


def test_forward_without_device_fails_loudly():
    from formula import synth_rays
    m = build()
    ro, rd, c2w = synth_rays(1, 4, 4)
    with pytest.raises(RuntimeError, match="no CPU"):
        m(ro, rd, c2w)
    with pytest.raises(RuntimeError, match="no CPU"):
        m.evaluate(ro, rd, c2w)


def test_schedules_follow_reference_trajectory():
    """lr sequence of torch's SequentialLR objects == closed form == reference golden lrs."""
    from papr_amd.schedule import lr_at
    g = golden("g7_trajectory.npz")
    m = build()
    cfg = case_cfg("chair1k")
    lr = cfg["training"]["lr"]
    key = {"points": "points", "attn": "attn", "points_influ_scores": "points_influ_scores", "pc_feats": "feats",
           "renderer": "generator"}
    for step in range(4):
        for name, opt in m.optimizers.items():
            want = lr_at(lr[key[name]], cfg["training"]["steps"], step)
            assert opt.param_groups[0]["lr"] == pytest.approx(want, rel=1e-9, abs=1e-30), (name, step)
        for p in m.parameters():
            if p.requires_grad:
                p.grad = torch.zeros_like(p)
        m.step(step)
    assert m.attn_lr == pytest.approx(lr_at(lr["attn"], cfg["training"]["steps"], 4), rel=1e-9)
    # after the reference's 3 steps its recorded lrs are those of scheduler epoch 3
    m2 = build()
    for step in range(3):
        for p in m2.parameters():
            if p.requires_grad:
                p.grad = torch.zeros_like(p)
        m2.step(step)
    assert m2.attn_lr == pytest.approx(float(g["attn_lr"]), rel=1e-9)
    assert m2.pts_lr == pytest.approx(float(g["pts_lr"]), rel=1e-9)


def test_init_optimizers_fast_forward_equals_stepping():
    """init_optimizers(n) jumps to step n in closed form; the state must be the one n scheduler steps leave behind,
    and stepping on from there must follow the same trajectory (below, inside and right at the end of the warm-up)."""
    import warnings
    cfg = case_cfg("chair1k")
    cfg["training"]["lr"]["attn"]["warmup"] = 40
    cfg["training"]["lr"]["points"]["warmup"] = 0
    cfg["training"]["steps"] = 300
    for n in (1, 25, 39, 40, 41, 120):
        m = build(cfg)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for step in range(n):
                for s_ in m.schedulers.values():
                    s_.step()
        m2 = build(cfg)
        m2.clear_optimizer(); m2.clear_scheduler()
        m2.init_optimizers(n)
        for extra in range(5):
            for name in m.optimizers:
                a_, b_ = m.optimizers[name].param_groups[0]["lr"], m2.optimizers[name].param_groups[0]["lr"]
                assert b_ == pytest.approx(a_, rel=1e-9, abs=1e-30), (n, extra, name)
                assert m2.schedulers[name].get_last_lr()[0] == pytest.approx(m.schedulers[name].get_last_lr()[0], rel=1e-9, abs=1e-30)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for s_ in list(m.schedulers.values()) + list(m2.schedulers.values()):
                    s_.step()

def test_prune_keeps_strictly_greater_and_drops_untouched_points():
    m = build()
    with torch.no_grad():
        m.points_influ_scores[:10] = 0.5
        m.points_influ_scores[10:20] = -0.1
    old_pts = m.points.data.clone()
    n = m.prune_points(0.0)           # influence init is exactly 0.0: `0 > 0` is false -> pruned
    assert int(n) == 990 and m.points.shape == (10, 3) and m.pc_feats.shape == (10, 64)
    assert torch.equal(m.points.data, old_pts[:10])
    assert isinstance(m.points, torch.nn.Parameter) and m.points.requires_grad


def test_add_points_matches_reference_algorithm():
    """Same numpy seed -> same new points as the reference's add_points_knn (restated in the test)."""
    from scipy.spatial import KDTree
    m = build()
    P0 = m.points.shape[0]
    pts = m.points.detach().clone().numpy()
    np.random.seed(7)
    n = m.add_points(50)
    assert n == 50 and m.points.shape[0] == P0 + 50 and m.points_influ_scores.shape == (P0 + 50, 1)
    # restate: sparsest sites by std of 10-NN distances, random convex combination of 3 neighbours
    np.random.seed(7)
    tree = KDTree(pts)
    d, _ = tree.query(pts, k=10)
    sites = np.argsort(d.std(axis=-1))[-50:]
    _, nn = tree.query(pts[sites], k=4)
    nn = nn[:, 1:]
    w = np.random.uniform(0, 1, (50, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    want = (pts[nn] * w[..., None]).sum(-2)
    np.testing.assert_allclose(m.points.detach().numpy()[P0:], want, rtol=0, atol=1e-6)
    assert torch.equal(m.points.detach()[:P0], torch.from_numpy(pts))


def test_save_load_roundtrip_with_changed_point_count(tmp_path):
    m = build()
    with torch.no_grad():
        m.points_influ_scores[:300] = 1.0
    m.prune_points(0.0)
    m.save(1234, str(tmp_path))
    for f in ("model.pth", "optimizers.pth", "schedulers.pth", "scaler.pth"):
        assert (tmp_path / f).exists()
    ck = torch.load(tmp_path / "model.pth")
    assert list(ck.keys()) == ["1234"]
    m2 = build(seed=5)
    assert m2.points.shape[0] == 1000
    step = m2.load(str(tmp_path))
    assert step == 1234 and m2.points.shape[0] == 300
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1


def test_config_merge_semantics():
    from papr_amd.config import load_config, deep_merge, eval_config, ConfigNode
    cfg = load_config("nerfsyn/lego.yml")
    assert cfg["geoms"]["background"]["constant"] == 3.0
    assert cfg["models"]["attn"]["embed"]["value"]["skip_layers"] == [5]
    assert cfg["models"]["attn"]["embed"]["key"]["ff_act"] == "leakyrelu"
    assert cfg["test"]["datasets"][0]["path"].endswith("lego/") and cfg["test"]["datasets"][0]["mode"] == "test"
    base = {"test": {"datasets": [{"name": "testset", "a": 1, "b": 2}]}}
    deep_merge(base, {"test": {"datasets": [{"name": "testset", "b": 3}, {"name": "extra", "a": 9}]}})
    assert base["test"]["datasets"] == [{"name": "testset", "a": 1, "b": 3}, {"name": "extra", "a": 9, "b": 3}]
    ev = eval_config(load_config("nerfsyn/chair.yml"))
    assert ev["dataset"]["mode"] == "test" and ev["dataset"]["extract_patch"] is False
    node = ConfigNode(cfg)
    assert node.geoms.points.select_k == 20 and "max_points" not in node


def test_unsupported_options_fail_at_construction():
    from papr_amd import get_model
    from papr_amd.config import deep_merge
    for over in ({"models": {"attn": {"embed": {"key": {"ff_act": "gelu"}}}}},
                 {"models": {"attn": {"embed": {"value": {"residual_layers": [2]}}}}},
                 {"exposure_control": {"use": True}},
                 {"models": {"renderer": {"generator": {"type": "big-unet"}}}}):
        cfg = deep_merge(case_cfg("chair1k"), over)
        with pytest.raises(NotImplementedError):
            get_model(cfg, device="cpu")
    with pytest.raises(KeyError, match="generator.mlp"):          # `type: mlp` needs its option block (default.yml ships none)
        get_model(deep_merge(case_cfg("chair1k"), {"models": {"renderer": {"generator": {"type": "mlp"}}}}), device="cpu")
    from papr_amd import get_loss
    with pytest.raises(NotImplementedError):
        get_loss({"mse": 1.0, "lpips_alex": 0.01})
    with pytest.raises(FileNotFoundError, match="vgg.pth"):          # LPIPS without its weight files: loud, with instructions
        cwd = os.getcwd()
        os.chdir(os.path.join(ROOT, "tests"))
        try:
            get_loss({"mse": 1.0, "lpips": 0.01})
        finally:
            os.chdir(cwd)


def test_lpips_loss_matches_reference_lpnet(tmp_path):
    """papr_amd.lpips.LPNet against the reference's LPNet (models/lpips.py:86-125) on a seeded image pair, backbone and heads
    formula-filled on both sides (golden G12); also through get_loss with weight files supplied the way a user would."""
    from formula import formula_fill
    from papr_amd import get_loss
    from papr_amd.lpips import LPNet
    g = golden("g12_lpips.npz")
    net = LPNet(load=False)
    formula_fill({"features." + k: v for k, v in net.features.state_dict().items()}, salt=3)
    with torch.no_grad():
        for i, p in enumerate(net.lins):
            p.copy_(torch.rand(p.shape, generator=torch.Generator().manual_seed(40 + i)))
    np.testing.assert_allclose([float(p.sum()) for p in net.lins], g["lin_sums"], rtol=1e-6)
    a = torch.from_numpy(g["a"]).requires_grad_(True)
    b = torch.from_numpy(g["b"])
    val = net(a, b)
    val.backward()
    assert abs(val.item() - float(g["value"])) <= 1e-6 * abs(float(g["value"]))
    assert abs(net(a[:1].detach(), b[:1]).item() - float(g["value_first"])) <= 1e-6 * abs(float(g["value_first"]))
    np.testing.assert_allclose(a.grad.numpy(), g["grad_a"], rtol=0, atol=1e-6 * np.abs(g["grad_a"]).max())
    # the user-facing route: weight files on disk -> get_loss builds mse + 0.01 lpips like the reference's get_loss
    torch.save({"lin%d.model.1.weight" % i: p.detach() for i, p in enumerate(net.lins)}, str(tmp_path / "vgg.pth"))
    torch.save({"features." + k: v for k, v in net.features.state_dict().items()}, str(tmp_path / "vgg16.pth"))
    os.environ["PAPR_LPIPS_HEADS"], os.environ["PAPR_VGG16_WEIGHTS"] = str(tmp_path / "vgg.pth"), str(tmp_path / "vgg16.pth")
    try:
        loss_fn = get_loss({"mse": 1.0, "lpips": 0.01, "lpips_alex": 0.0})
    finally:
        del os.environ["PAPR_LPIPS_HEADS"], os.environ["PAPR_VGG16_WEIGHTS"]
    want = torch.mean((a.detach() - b) ** 2).item() + 0.01 * float(g["value"])
    assert abs(loss_fn(a.detach(), b).item() - want) <= 1e-6


def test_c_abi_library_exports_every_declared_symbol():
    from papr_amd import hip
    header = open(os.path.join(ROOT, "include", "papr_hip.h")).read()
    declared = set(re.findall(r"\b(papr_[a-z0-9_]+)\s*\(", header))
    assert declared == set(hip.EXPORTS), declared ^ set(hip.EXPORTS)
    assert os.path.exists(hip.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.papr_abi_version.restype = ctypes.c_int
    assert lib.papr_abi_version() == hip.EXPECTED_ABI
    lib.papr_mlp_bwd_workspace_bytes.restype = ctypes.c_size_t
    lib.papr_mlp_bwd_workspace_bytes.argtypes = [ctypes.c_int64]
    lib.papr_mlp_fwd_workspace_bytes.restype = ctypes.c_size_t
    lib.papr_mlp_fwd_workspace_bytes.argtypes = [ctypes.c_int64]
    assert lib.papr_mlp_bwd_workspace_bytes(1000) >= 256 * (256 * 256 + 256) * 4 + lib.papr_mlp_fwd_workspace_bytes(1000)
    assert lib.papr_mlp_fwd_workspace_bytes(1000) >= 2 * 1000 * 4 + 2 * 512 * 704 * 2


def test_c_abi_round5_entry_points_validate_their_arguments_before_touching_a_device():
    """The ABI-26 entry points (whole-network U-Net, score-bias gradient, batched folds, MSE) refuse bad shapes and null pointers with a message and
    a non-zero status; the U-Net's size queries are consistent.  No device call is reached: this runs on the CPU box."""
    from papr_amd import hip
    lib = ctypes.CDLL(hip.LIB_PATH)
    lib.papr_last_error.restype = ctypes.c_char_p
    sz = lib.papr_small_unet_state_bytes
    sz.restype, sz.argtypes = ctypes.c_size_t, [ctypes.c_int32] * 5
    keep, lean = sz(1, 160, 160, 32, 1), sz(1, 160, 160, 32, 0)
    acts = 4 * (25600 * 256 + 6400 * 512 + 6400 * 128 + 1600 * 256 + 1600 * 512 + 6400 * 256 + 25600 * 128)      # cat2, cat1, pool1, pool2, x3, y1, y2
    assert lean >= acts and keep > lean and sz(1, 320, 320, 32, 1) > 2 * keep
    bw = lib.papr_small_unet_bwd_workspace_bytes
    bw.restype, bw.argtypes = ctypes.c_size_t, [ctypes.c_int32] * 5
    assert bw(1, 160, 160, 32, 3) >= 4 * (25600 * (128 + 256 + 128) + 6400 * (256 + 512 + 256 + 128) + 1600 * (512 + 256))
    d = hip.UnetDesc()
    d.B, d.H, d.W, d.c_in, d.n_classes = 1, 30, 40, 32, 3
    fwd = lib.papr_small_unet_fwd
    fwd.argtypes = [ctypes.POINTER(hip.UnetDesc)] + [ctypes.c_void_p] * 3 + [ctypes.c_int32, ctypes.c_void_p]
    assert fwd(ctypes.byref(d), None, None, None, 1, None) != 0 and b"multiples of 4" in lib.papr_last_error()
    d.H = 40
    d.c_in = 48
    assert fwd(ctypes.byref(d), None, None, None, 1, None) != 0 and b"multiple of 32" in lib.papr_last_error()
    d.c_in = 32
    assert fwd(ctypes.byref(d), None, None, None, 1, None) != 0 and b"null weight" in lib.papr_last_error()
    qk = lib.papr_qk_bias_bwd
    qk.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 9
    assert qk(*([None, 64, 64, 256, None, 10, None, 64] + [None] * 9)) != 0 and b"null pointer" in lib.papr_last_error()
    fb = lib.papr_ln_fold_fwd_batch
    fb.argtypes = [ctypes.POINTER(hip.LnFoldJob), ctypes.c_int32, ctypes.c_void_p]
    jobs = (hip.LnFoldJob * 1)()
    assert fb(jobs, 0, None) != 0 and fb(jobs, hip.LN_FOLD_MAX_JOBS + 1, None) != 0 and b"jobs" in lib.papr_last_error()
    assert fb(jobs, 1, None) != 0 and b"null pointer" in lib.papr_last_error()
    mse = lib.papr_mse_fwd
    mse.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    assert mse(None, None, 10, None, None, None, None) != 0 and b"papr_mse_fwd" in lib.papr_last_error()


def test_generated_kernel_sources_are_current():
    """chain4_kloop.inc / chain4_fused.inc are what their generators print (a stale include would still build -- and run another kernel
    than the scripts describe)."""
    import subprocess
    import sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for script, inc in (("gen_chain4_kloop.py", "chain4_kloop.inc"), ("gen_chain4_fused.py", "chain4_fused.inc")):
        env = {k: v for k, v in os.environ.items() if not k.startswith("C4F_")}          # (the generators' experiment switches)
        out = subprocess.run([_sys.executable, os.path.join(root, "scripts", script)], capture_output=True, text=True, env=env, check=True).stdout
        assert out == open(os.path.join(root, "papr_amd", "csrc", inc)).read(), "%s is not what scripts/%s prints" % (inc, script)


def test_mlp_generator_surface():
    """The per-pixel MLP render head (reference models/renderer.py:6-17) keeps the reference's parameter names, refuses
    what the kernels cannot do, and -- like the rest of the path -- has no CPU fallback."""
    from papr_amd.unet import MLPGenerator, get_generator
    seed_all(3)
    m = MLPGenerator(32, 3, 128, 3, act_type="leakyrelu", last_act_type="none")
    assert sorted(m.state_dict().keys()) == ["mlp.model.%d.%s" % (i, k) for i in (1, 3, 5) for k in ("bias", "weight")]
    assert m.state_dict()["mlp.model.1.weight"].shape == (128, 32) and m.state_dict()["mlp.model.5.weight"].shape == (3, 128)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 32, 4, 4))
    opt = dict(num_layers=3, num_channels=64, act_type="relu", last_act_type="none", use_wn=False, act_a=1.0, act_b=1.0,
               act_trainable=False, skip_layers=[1], bias=True, half_layers=[], residual_layers=[], residual_dims=[])
    g = get_generator({"type": "mlp", "mlp": opt}, 32, 3)
    assert g.state_dict()["mlp.model.3.weight"].shape == (64, 64 + 32)          # skip layer: [hidden | input]
    for bad in (dict(use_wn=True), dict(bias=False), dict(half_layers=[1]), dict(act_type="gelu"), dict(act_a=2.0)):
        with pytest.raises(NotImplementedError):
            get_generator({"type": "mlp", "mlp": dict(opt, **bad)}, 32, 3)


def test_ssim_restatement_properties():
    """papr_amd/metrics.py: SSIM as the reference calls it (uniform 11 x 11 window, sample covariance, data_range 1)."""
    from papr_amd.metrics import psnr, ssim
    rng = np.random.default_rng(0)
    a = rng.random((40, 37, 3))
    assert ssim(a, a) == pytest.approx(1.0, abs=1e-12)
    b = np.clip(a + 0.05 * rng.standard_normal(a.shape), 0, 1)
    s_ab = ssim(a, b)
    assert 0.0 < s_ab < 1.0 and s_ab == pytest.approx(ssim(b, a), abs=1e-12)          # symmetric
    assert ssim(a, np.clip(a + 0.2 * rng.standard_normal(a.shape), 0, 1)) < s_ab           # more noise, less similar
    # two constant images: means only.  SSIM = (2 x y + C1) / (x^2 + y^2 + C1) with C1 = 1e-4 (variances are exactly 0)
    x, y = np.full((20, 20, 1), 0.3), np.full((20, 20, 1), 0.6)
    assert ssim(x, y) == pytest.approx((2 * 0.3 * 0.6 + 1e-4) / (0.09 + 0.36 + 1e-4), rel=1e-9)
    assert psnr(x, y) == pytest.approx(-10 * np.log10(0.09), rel=1e-12)


def test_loads_the_checkpoint_directory_the_reference_wrote():
    """tests/golden/g13_ref_ckpt: model.pth / optimizers.pth / schedulers.pth / scaler.pth as the reference's PAPR.save
    (models/model.py:562-586) left them after three of its own train steps.  PAPR.load(dir, load_optimizer=True) must take all of it:
    every state-dict entry bit for bit, the Adam moments and step counters, the scheduler positions."""
    from conftest import G13_DIR, g13_cfg
    from papr_amd import get_model
    seed_all(9)
    m = get_model(g13_cfg(), device="cpu")
    step = m.load(G13_DIR, load_optimizer=True)
    assert step == 3
    ref = torch.load(os.path.join(G13_DIR, "model.pth"), map_location="cpu")["3"]
    own = m.state_dict()
    assert list(own.keys()) == list(ref.keys())
    for k, v in ref.items():
        assert torch.equal(own[k], v), k
    osd = torch.load(os.path.join(G13_DIR, "optimizers.pth"), map_location="cpu")
    assert set(osd) == set(m.optimizers)
    for name, opt in m.optimizers.items():
        got = opt.state_dict()
        assert len(got["state"]) == len(osd[name]["state"]) > 0, name
        for i, st in osd[name]["state"].items():
            assert float(got["state"][i]["step"]) == 3.0
            assert torch.equal(got["state"][i]["exp_avg"], st["exp_avg"]) and torch.equal(got["state"][i]["exp_avg_sq"], st["exp_avg_sq"])
    ssd = torch.load(os.path.join(G13_DIR, "schedulers.pth"), map_location="cpu")
    for name, sch in m.schedulers.items():
        assert sch.state_dict()["last_epoch"] == ssd[name]["last_epoch"] == 3, name


def test_test_time_lpips_is_the_pinned_network_without_input_rescaling():
    """test.py's LPIPS (papr_amd/lpips.py: TestLPIPS, the `lpips` package's v0.1 metric as reference test.py:109-110 calls it) on the VGG
    arm is the arithmetic of the training loss's network -- which G12 pins against the reference's LPNet -- minus the 2x-1 input
    rescaling (the reference hands [0,1] images to the package without normalize=True); the AlexNet arm has the package's tap layout."""
    from papr_amd.lpips import LPNet, TestLPIPS
    torch.manual_seed(0)
    ln = LPNet(load=False)
    for p in ln.features.parameters():
        p.data.normal_(0, 0.05)
    for p in ln.lins:
        p.data.uniform_(0, 1)
    t = TestLPIPS("vgg")
    t.features.load_state_dict(ln.features.state_dict())
    for a, b in zip(t.lins, ln.lins):
        a.data.copy_(b.data)
    x, y = torch.rand(2, 32, 32, 3), torch.rand(2, 32, 32, 3)
    got = t((2 * x - 1).permute(0, 3, 1, 2), (2 * y - 1).permute(0, 3, 1, 2))
    assert got.shape == (2,)
    assert abs(float(got.mean()) - float(ln(x, y).detach())) < 1e-6
    assert float(t(x.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2)).abs().max()) == 0.0
    al = TestLPIPS("alex")
    assert [al.features[i].out_channels for i in (0, 3, 6, 8, 10)] == [64, 192, 384, 256, 256] and al.taps == (1, 4, 7, 9, 11)
    assert TestLPIPS.try_build("alex", "cpu") is None       # no weights in this environment: test.py reports nan


def test_depth_map_is_attention_weighted_plane_distance():
    from papr_amd.metrics import depth_map
    g = torch.Generator().manual_seed(3)
    o = torch.tensor([[0.0, 0.0, 40.0]])
    sel = torch.randn(1, 5, 6, 4, 3, generator=g) * 3
    attn = torch.softmax(torch.randn(1, 5, 6, 5, 1, generator=g), dim=-2)
    d = depth_map(sel, attn, o)
    want = (attn[0, ..., :4, 0] * (40.0 - sel[0, ..., 2]).abs()).sum(-1).numpy()     # camera on the z axis: distance to the plane z = 40
    assert d.shape == (5, 6) and np.allclose(d, want, atol=1e-5)


def test_output_activation_matches_the_reference_for_every_name():
    """models.last_act (reference activation_func, models/utils.py:183-229, default arguments): values and the state-dict keys each choice adds."""
    from papr_amd.activations import output_activation
    g = golden("g16_last_act.npz")
    x = torch.from_numpy(g["x"])
    names = [k[2:] for k in g.files if k.startswith("y/")]
    assert len(names) == 18
    for name in names:
        layer = output_activation(name)
        np.testing.assert_allclose(layer(x.clone()).detach().numpy(), g["y/" + name], rtol=0, atol=1e-7, err_msg=name)
        assert sorted(layer.state_dict().keys()) == [str(k) for k in g["keys/" + name]], name
    with pytest.raises(NotImplementedError):
        output_activation("no-such-activation")
    import copy
    cfg = copy.deepcopy(case_cfg("chair1k"))
    cfg["models"]["last_act"] = "gaussian"
    from papr_amd import get_model
    m = get_model(cfg, device="cpu")
    assert "last_act.a" in m.state_dict() and float(m.last_act(torch.zeros(1))) == 1.0


def test_eval_chunk_policy(monkeypatch):
    """train.eval_chunk: the drivers' chunk of an image per evaluate() call -- the configured size on the CPU and under PAPR_EVAL_CHUNK=config, grown
    towards 160,000 rays (never beyond the image, never beyond a quarter of the free device memory) otherwise."""
    import types
    import train
    m = types.SimpleNamespace(points=torch.zeros(30000, 3), select_k=20)
    cuda = types.SimpleNamespace(type="cuda")
    assert train.eval_chunk(m, 1, 800, 800, 200, 200, torch.device("cpu")) == (200, 200)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d: (200 << 30, 288 << 30))
    monkeypatch.delenv("PAPR_EVAL_CHUNK", raising=False)
    assert train.eval_chunk(m, 1, 800, 800, 200, 200, cuda) == (400, 400)
    assert train.eval_chunk(m, 1, 800, 800, 100, 100, cuda) == (400, 400)
    assert train.eval_chunk(m, 1, 100, 100, 200, 200, cuda) == (100, 100)          # never beyond the image
    assert train.eval_chunk(m, 1, 1080, 1920, 200, 200, cuda) == (400, 400)
    h, w = train.eval_chunk(m, 1, 800, 800, 50, 800, cuda)
    assert h * w <= train.EVAL_CHUNK_RAYS and (h, w) == (200, 800)
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda d: (8 << 30, 288 << 30))   # little memory left: the chunk stops growing
    h, w = train.eval_chunk(m, 1, 800, 800, 100, 100, cuda)
    assert h * w * 4 * 20 * 1024 <= (8 << 30) // 4 and (h, w) != (400, 400)
    monkeypatch.setenv("PAPR_EVAL_CHUNK", "config")
    assert train.eval_chunk(m, 1, 800, 800, 200, 200, cuda) == (200, 200)



class _FakeDeviceMap:
    """What the render head's routing looks at of a map: device, dtype, shape (no GPU in the CPU suite)."""
    is_cuda = True

    def __init__(self, shape, dtype=torch.float32):
        self.shape, self.dtype = tuple(shape), dtype


def test_render_head_has_no_torch_fallback_on_the_device(monkeypatch, capsys):
    """papr_amd/unet.py: a DEVICE tensor can reach torch's own layers (nn.Conv2d, MaxPool2d, ConvTranspose2d, autocast) only behind the one debug switch
    PAPR_DEBUG_TORCH_HEAD, which says so once on stderr; a shape the own kernels do not cover raises by name (rounds 3-5: four import-time switches and
    silent shape tests).  Checked on the routing function every layer goes through, with a stand-in for a device map, and on the module source: every call
    of a torch layer sits behind that function."""
    import ast
    import inspect
    import papr_amd.debug as dbg
    import papr_amd.unet as unet
    monkeypatch.setattr(dbg, "_ON", frozenset())
    monkeypatch.setattr(dbg, "_said", set())
    ok_map = _FakeDeviceMap((1, 32, 8, 8))
    assert unet._own(ok_map, True, "conv", "Conv2d(32, 128, 3x3)", "channels") is True             # the own kernel
    for bad in (dict(ok=False), dict(ok=True, dtype=torch.float16), dict(ok=True, dtype=torch.float64)):
        with pytest.raises(NotImplementedError, match="no torch fallback on the device"):
            unet._own(_FakeDeviceMap((1, 32, 8, 8), bad.get("dtype", torch.float32)), bad["ok"], "rest", "MaxPool2d(2)", "channels")
    with monkeypatch.context() as mp:                                                                # a caller's (device) autocast region: not silently honoured either
        mp.setattr(torch, "is_autocast_enabled", lambda *a: True)
        with pytest.raises(NotImplementedError, match="under autocast"):
            unet._own(ok_map, True, "conv", "Conv2d(32, 128, 3x3)", "channels")
    assert unet._own(torch.zeros(1, 32, 8, 8), True, "conv", "Conv2d", "channels") is False          # a CPU tensor: plain torch, the tests' reference
    assert capsys.readouterr().err == ""
    # the debug switch: torch, and it says so -- once per part
    monkeypatch.setattr(dbg, "_ON", frozenset({"rest"}))
    assert unet._own(ok_map, True, "rest", "MaxPool2d(2)", "channels") is False
    assert unet._own(ok_map, False, "rest", "MaxPool2d(2)", "channels") is False
    assert unet._own(ok_map, True, "conv", "Conv2d(32, 128, 3x3)", "channels") is True               # (another part stays on the own kernels)
    err = capsys.readouterr().err
    assert err.count("PAPR_DEBUG_TORCH_HEAD") == 1 and "'rest'" in err
    # the source: a torch layer of the head is only ever called in the branch behind _own / _own_rest, and no other environment switch routes layers
    src = inspect.getsource(unet)
    tree = ast.parse(src)
    for cls in ("ConvStage", "DownStage", "UpStage", "Head"):
        fwd = next(f for c in tree.body if isinstance(c, ast.ClassDef) and c.name == cls for f in c.body if isinstance(f, ast.FunctionDef) and f.name == "forward")
        tests = [n for n in ast.walk(fwd) if isinstance(n, ast.If)]
        assert any(isinstance(t.test, ast.Call) and getattr(t.test.func, "id", "") in ("_own", "_own_rest") for t in tests), cls
    assert "PAPR_UNET_CONV" not in src and "PAPR_UNET_REST" not in src and 'get("PAPR_UNET_AMP"' not in src
    with pytest.raises(ValueError):                                                                  # an unknown part is an error, not a no-op
        monkeypatch.setenv("PAPR_DEBUG_TORCH_HEAD", "convs")
        import importlib
        importlib.reload(dbg)
    monkeypatch.delenv("PAPR_DEBUG_TORCH_HEAD")
    importlib.reload(dbg)


def test_small_unet_last_act_option():
    """`models.renderer.generator.small_unet.last_act` (reference models/unet.py:205,253: `activation_func(last_act)` on the logits): any name of the
    table the model's own output activation uses (G16) -- the head's result through that function, the reference's state-dict keys (`renderer.last_act.a`
    for the names with a shape constant, none for `none`); the other non-shipped variants (bilinear, double conv, norm, affine) still raise by name."""
    import torch
    from papr_amd import get_model
    from papr_amd.config import deep_merge
    from papr_amd.unet import SmallUNet
    gen = lambda o: {"models": {"renderer": {"generator": {"small_unet": o}}}}
    m = get_model(deep_merge(case_cfg("chair1k"), gen({"last_act": "sigmoid"})), device="cpu")
    assert not [k for k in m.state_dict() if k.startswith("renderer.last_act")]
    m2 = get_model(deep_merge(case_cfg("chair1k"), gen({"last_act": "gaussian"})), device="cpu")
    assert "renderer.last_act.a" in m2.state_dict()
    torch.manual_seed(0)
    plain, act = SmallUNet(32, 3), SmallUNet(32, 3, last_act="sigmoid")
    act.load_state_dict(plain.state_dict())
    x = torch.randn(1, 32, 8, 12)
    assert torch.equal(act(x), torch.sigmoid(plain(x)))
    for o in ({"bilinear": True}, {"single": False}, {"norm": "batch"}, {"affine_layer": 1}):
        with pytest.raises(NotImplementedError, match="small-unet"):
            get_model(deep_merge(case_cfg("chair1k"), gen(o)), device="cpu")
