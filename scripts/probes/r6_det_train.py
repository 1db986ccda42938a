"""Run-to-run determinism of training steps at full size: python3 scripts/probes/r6_det_train.py <scene> <steps> <out.pt>   (scene: nerfsyn/lego.yml ...)
Saves a float64 sum and an int64 bit checksum of every parameter's gradient at every step; scripts/probes/r6_det_train.sh compares two processes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from papr_amd import get_loss, get_model, load_config
from papr_amd.data import SyntheticRayData
scene, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
cfg = load_config(scene, overrides={"use_amp": os.environ.get("DET_AMP", "0") == "1", "training": {"losses": {"mse": 1.0, "lpips": 0.0, "lpips_alex": 0.0}}})
torch.manual_seed(1); np.random.seed(1)
m = get_model(cfg, device="cuda").to("cuda")
with torch.no_grad():
    m.points_influ_scores.uniform_(0.0, 1.0)
m.clear_optimizer(); m.clear_scheduler(); m.init_optimizers(0)
data = SyntheticRayData(cfg["dataset"], n_views=4, seed=3, device="cuda")
loss_fn = get_loss(cfg["training"]["losses"]).to("cuda")
log = []
for step in range(steps):
    tgt, rayd, rayo, c2w = data.patch()
    m.clear_grad()
    o = m(rayo, rayd, c2w, step)
    loss = loss_fn(m.last_act(o) if hasattr(m, "last_act") else o, tgt)
    m.scaler.scale(loss).backward()
    rec = {"loss": float(loss)}
    for n, p in m.named_parameters():
        if p.grad is not None:
            rec[n] = (float(p.grad.double().sum()), int(p.grad.view(torch.int32).long().sum()))
    log.append(rec)
    m.step(step); m.scaler.update()
torch.save(log, out)
