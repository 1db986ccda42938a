#!/usr/bin/env python3
"""Full-image inference (BASELINE config 3): lego hyper-parameters, P = 30,000 points, one 800x800 view
rendered through PAPR.evaluate in chunks + U-Net + compositing (the reference's test_step loop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from papr_amd import get_model, load_config
from papr_amd.data import SyntheticRayData
from train import render_full

P = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 200
os.environ["PAPR_EVAL_CHUNK"] = "config"            # (the chunk named on the command line, not the drivers' choice: train.py, eval_chunk)
cfg = load_config("nerfsyn/lego.yml", overrides={"use_amp": False, "geoms": {"points": {"init_num": P}},
                                                 "training": {"losses": {"mse": 1.0, "lpips": 0.0}}})
torch.manual_seed(1); np.random.seed(1)
so = sys.stdout; sys.stdout = open(os.devnull, "w")
m = get_model(cfg, "cpu"); sys.stdout = so
with torch.no_grad():
    m.points_influ_scores.uniform_(0, 1)
m = m.to("cuda")
data = SyntheticRayData(cfg["dataset"], n_views=4, seed=0, device="cuda")
img, rayd, rayo, c2w = data.full_view(0)
for _ in range(2):
    render_full(m, rayo, rayd, c2w, chunk, chunk)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 3
for _ in range(n):
    rgb = render_full(m, rayo, rayd, c2w, chunk, chunk)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("render 800x800, P=%d, chunk %dx%d: %.1f ms/image, %.2f M rays/s" % (P, chunk, chunk, dt * 1e3, 0.64 / dt))
if len(sys.argv) > 3:       # a JSON record for profiles/
    import json
    json.dump({"workload": "configs/nerfsyn/lego.yml (value MLP with skip layer 5, LeakyReLU), P=%d points, one 800x800 view = 640,000 rays through PAPR.evaluate in "
                           "%dx%d chunks + SmallUNet + compositing (reference test_step loop, test.py:59-104); fp32 parity mode; data resident" % (P, chunk, chunk),
               "ms_per_image": dt * 1e3, "rays_per_s": 0.64e6 / dt, "images_timed": n, "warmup_images": 2}, open(sys.argv[3], "w"), indent=1)
