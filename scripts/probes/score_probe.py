import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from papr_amd import get_model, load_config
from papr_amd.data import SyntheticRayData
for where in ("cuda", "cpu"):
    cfg = load_config("nerfsyn/chair.yml", overrides={"use_amp": False, "training": {"losses": {"mse": 1.0, "lpips": 0.0, "lpips_alex": 0.0}}})
    torch.manual_seed(1); np.random.seed(1)
    m = get_model(cfg, device=where).to("cuda")
    with torch.no_grad():
        m.points_influ_scores.uniform_(0.0, 1.0)
    data = SyntheticRayData(cfg["dataset"], n_views=4, seed=3, device="cuda")
    for i in range(3):
        tgt, rayd, rayo, c2w = data.patch()
        with torch.no_grad():
            fused, attn = m.evaluate(rayo, rayd, c2w)
        a = attn[..., :-1, 0]
        print(where, "patch", i, "chain", os.environ.get("PAPR_CHAIN", "2"), "fg attn min %.6f max %.6f; white frac %.2f; |fused| max %.3e; influ max %.3f" %
              (float(a.min()), float(a.max()), float((tgt > 0.99).float().mean()), float(fused.abs().max()), float(m.points_influ_scores.max())))
