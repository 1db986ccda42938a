"""Is the training step CPU-bound?  Time to ENQUEUE n steps (no synchronisation) against the time until the GPU has finished them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from papr_amd import get_model, get_loss, load_config
from papr_amd.data import SyntheticRayData
cfg = load_config("nerfsyn/chair.yml", overrides={"use_amp": False, "training": {"losses": {"mse": 1.0, "lpips": 0.0, "lpips_alex": 0.0}}})
torch.manual_seed(1); np.random.seed(1)
so = sys.stdout; sys.stdout = open(os.devnull, "w")
m = get_model(cfg, "cuda").to("cuda"); sys.stdout = so
with torch.no_grad():
    m.points_influ_scores.uniform_(0, 1)
loss_fn = get_loss(cfg["training"]["losses"]).to("cuda")
data = SyntheticRayData(cfg["dataset"], n_views=4, seed=0, device="cuda")
batch = data.patch()
def step(i):
    tgt, rayd, rayo, c2w = batch
    m.clear_grad()
    out = m.last_act(m(rayo, rayd, c2w, i))
    loss = loss_fn(out, tgt)
    m.scaler.scale(loss).backward()
    m.step(i)
    m.scaler.update()
for i in range(5): step(i)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for i in range(n): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("20 steps back to back: enqueue %.2f ms/step, finished %.2f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
ts = []
for i in range(10):                      # one step from an idle queue: how long the CPU needs to enqueue it
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(i); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("single step: enqueue %.2f ms, finished %.2f ms" % (np.median([a for a, b in ts]) * 1e3, np.median([b for a, b in ts]) * 1e3))
