import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from papr_amd import get_loss, get_model, load_config
from papr_amd.data import SyntheticRayData
for amp in (False, True):
    cfg = load_config("nerfsyn/chair.yml", overrides={"use_amp": amp, "training": {"losses": {"mse": 1.0, "lpips": 0.0, "lpips_alex": 0.0}}})
    torch.manual_seed(1); np.random.seed(1)
    m = get_model(cfg, device="cuda").to("cuda")
    with torch.no_grad():
        m.points_influ_scores.uniform_(0.0, 1.0)
    data = SyntheticRayData(cfg["dataset"], n_views=4, seed=3, device="cuda")
    loss_fn = get_loss(cfg["training"]["losses"]).to("cuda")
    tgt, rayd, rayo, c2w = data.patch()
    m.clear_grad()
    out = m(rayo, rayd, c2w, 0)
    loss = loss_fn(out, tgt)
    m.scaler.scale(loss).backward()
    for n in ("points", "pc_feats", "points_influ_scores", "proximity_attn.attention_layer.w_q.weight"):
        g = dict(m.named_parameters())[n].grad
        print(amp, n, float(g.abs().max()), int((g != 0).sum()), g.numel())
    fused, attn = m.evaluate(rayo, rayd, c2w)
    print("attn bkg mean", float(attn[..., -1, 0].mean()), "attn fg max", float(attn[..., :-1, 0].max()), "target white frac", float((tgt > 0.99).float().mean()))
