import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm device (MI355X); run with -m gpu")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no ROCm device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


SMALL = {"geoms": {"points": {"init_num": 1000}}}
TINY = {"geoms": {"points": {"init_num": 1000, "select_k": 12}},
        "models": {"use_renderer": False, "attn": {"d_model": 64, "embed": {
            "k_L": [4, 4, 4], "q_L": [4], "v_L": [4, 4],
            "key": {"d_ff": 64, "d_ff_out": 64, "n_ff_layer": 3},
            "query": {"d_ff": 64, "d_ff_out": 64, "n_ff_layer": 3},
            "value": {"d_ff": 64, "d_ff_out": 3, "n_ff_layer": 4}}}}}
PARITY = {"use_amp": False, "training": {"losses": {"mse": 1.0, "lpips": 0.0, "lpips_alex": 0.0}}}

VARIANTS = {"geoms": {"points": {"init_num": 1000}, "point_feats": {"use_ink": True, "use_inv": True}},
            "models": {"normalize_topk_attn": False, "attn": {"embed": {"embed_type": 2}}}}

# G13: the tiny model whose checkpoint directory the REFERENCE's own PAPR.save wrote (tests/golden/g13_ref_ckpt, make_golden.py --round3)
G13 = {"geoms": {"points": {"init_num": 400, "select_k": 12}, "point_feats": {"dim": 16}},
       "models": {"use_renderer": False, "attn": {"d_model": 64, "embed": {
           "k_L": [4, 4, 4], "q_L": [4], "v_L": [4, 4],
           "key": {"d_ff": 64, "d_ff_out": 64, "n_ff_layer": 3},
           "query": {"d_ff": 64, "d_ff_out": 64, "n_ff_layer": 3},
           "value": {"d_ff": 64, "d_ff_out": 3, "n_ff_layer": 4}}}}}
G13_DIR = os.path.join(GOLDEN, "g13_ref_ckpt")


def g13_cfg():
    from papr_amd.config import load_config, deep_merge
    return deep_merge(load_config("nerfsyn/chair.yml", overrides=G13), PARITY)


# weight-normalised embedding MLPs (`use_wn: true`; no shipped scene file sets it): G13's tiny model with 1,000 points
WN_TINY = {"geoms": {"points": {"init_num": 1000, "select_k": 12}, "point_feats": {"dim": 16}},
           "models": {"use_renderer": False, "attn": {"d_model": 64, "embed": {
               "k_L": [4, 4, 4], "q_L": [4], "v_L": [4, 4],
               "key": {"d_ff": 64, "d_ff_out": 64, "n_ff_layer": 3, "use_wn": True},
               "query": {"d_ff": 64, "d_ff_out": 64, "n_ff_layer": 3, "use_wn": True},
               "value": {"d_ff": 64, "d_ff_out": 3, "n_ff_layer": 4, "use_wn": True}}}}}

# half layers (`half_layers`; no shipped scene file sets them): the key MLP's layer 1 and the value MLP's layer 2 take half the width
HALF_TINY = {"geoms": {"points": {"init_num": 1000, "select_k": 12}, "point_feats": {"dim": 16}},
             "models": {"use_renderer": False, "attn": {"d_model": 64, "embed": {
                 "k_L": [4, 4, 4], "q_L": [4], "v_L": [4, 4],
                 "key": {"d_ff": 64, "d_ff_out": 64, "n_ff_layer": 3, "half_layers": [1]},
                 "query": {"d_ff": 64, "d_ff_out": 64, "n_ff_layer": 3},
                 "value": {"d_ff": 64, "d_ff_out": 3, "n_ff_layer": 4, "half_layers": [2]}}}}}

CASES = {
    "chair1k": ("nerfsyn/chair.yml", SMALL, dict(n_img=1, hw=16, seed=0)),
    "lego1k": ("nerfsyn/lego.yml", SMALL, dict(n_img=1, hw=16, seed=0)),
    "tiny_norender": ("nerfsyn/chair.yml", TINY, dict(n_img=2, hw=8, seed=4)),
    # feature variants no shipped scene file uses: point features in the key as well (use_ink), posenc without the raw
    # coordinate (embed_type 2), un-normalised top-k attention
    "variants1k": ("nerfsyn/chair.yml", VARIANTS, dict(n_img=1, hw=16, seed=0)),
    "wn_tiny": ("nerfsyn/chair.yml", WN_TINY, dict(n_img=2, hw=8, seed=4)),
    "half_tiny": ("nerfsyn/chair.yml", HALF_TINY, dict(n_img=2, hw=8, seed=4)),
}


def case_cfg(tag):
    from papr_amd.config import load_config, deep_merge
    scene, over, _ = CASES[tag]
    cfg = load_config(scene, overrides=over)
    return deep_merge(cfg, PARITY)


def case_rays(tag):
    from formula import synth_rays
    r = CASES[tag][2]
    return synth_rays(r["n_img"], r["hw"], r["hw"], seed=r["seed"])


def grad_check(got, ref, name=""):
    """Gradient parity bar.  e = |got - ref| / max |ref| over the tensor:
      * rms(e) <= 1.5e-4 and e <= 1e-3 for all but 0.1 % of the elements (measured on every golden case: rms 1e-8 ... 1.2e-4, typical
        maximum 1e-6 ... 7e-4: fp32 sums of 5,120 - 512,000 terms in another order, 22-bit split-f16 operands);
      * the remaining elements <= 1e-2: isolated outliers -- a ReLU / LeakyReLU pre-activation within rounding of zero takes the
        other branch of the derivative than in the reference's own fp32 evaluation, which changes ONE pair's gradient row by
        a fraction of a percent.  `scripts/probes/grad_err.py` shows them: a few dozen of 64,000 elements, in different
        cases for the exact-fp32 MFMA mode (chair1k) and the split-f16 mode (lego1k), rms unchanged."""
    import numpy as np
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    scale = max(np.abs(ref).max(), 1e-30)
    e = np.abs(got - ref) / scale
    rms, frac, worst = float(np.sqrt((e ** 2).mean())), float((e > 1e-3).mean()), float(e.max())
    assert rms <= 1.5e-4 and frac <= 1e-3 and worst <= 1e-2, (name, rms, frac, worst)
    return worst
