"""Which torch ops launch the small kernels of a training step (torch.profiler, one step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from papr_amd import get_model, get_loss, load_config
from papr_amd.data import SyntheticRayData
cfg = load_config("nerfsyn/chair.yml", overrides={"use_amp": False, "training": {"losses": {"mse": 1.0, "lpips": 0.0, "lpips_alex": 0.0}}})
torch.manual_seed(1); np.random.seed(1)
so = sys.stdout; sys.stdout = open(os.devnull, "w")
m = get_model(cfg, "cpu"); sys.stdout = so
m = m.to("cuda")
m.clear_optimizer(); m.clear_scheduler(); sys.stdout = open(os.devnull, "w"); m.init_optimizers(0); sys.stdout = so
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
for i in range(3): step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    step(3); torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
print("%-60s %6s %10s" % ("op", "count", "cuda_us"))
for e in rows[:70]:
    print("%-60s %6d %10.1f" % (e.key[:60], e.count, getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0.0))))
