"""Which aten ops the training step still launches (the torch glue around the C-ABI calls): bench.py's step under torch.profiler,
device time per op name over 5 steps."""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
sys.argv = ["bench.py"]
args = bench.parse()
os.environ["PAPR_GEMM_MODE"] = "h3"
from papr_amd import get_model, get_loss
from papr_amd.data import SyntheticRayData
dev = torch.device("cuda", 0)
cfg = bench.bench_config(args.scene, args.points, args.amp)
torch.manual_seed(cfg["seed"])
model = get_model(cfg, device="cpu").to(dev)
model.clear_optimizer(); model.clear_scheduler(); model.init_optimizers(0)
loss_fn = get_loss(cfg["training"]["losses"]).to(dev)
data = SyntheticRayData(cfg["dataset"], n_views=100, seed=100, device=dev)
pool = [data.patch() for _ in range(4)]
for i in range(4):
    bench.train_step(model, loss_fn, pool[i % 4], i)
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=bool(os.environ.get("STACKS"))) as prof:
    for i in range(5):
        bench.train_step(model, loss_fn, pool[i % 4], 4 + i)
    torch.cuda.synchronize()
if os.environ.get("STACKS"):                      # STACKS=1: per (op, first frame inside this repo), so that every launch has an address
    rows = {}
    for e in prof.key_averages(group_by_stack_n=12):
        if e.self_device_time_total <= 0 or not e.key.startswith("aten::"):
            continue
        here = [f for f in e.stack if "/papr_amd/" in f or "bench.py" in f or "train.py" in f]
        site = here[0].split("/")[-1] if here else (e.stack[0] if e.stack else "?")
        k = (e.key, site)
        c, t = rows.get(k, (0, 0))
        rows[k] = (c + e.count, t + e.self_device_time_total)
    out = sorted(rows.items(), key=lambda r: -r[1][1])
    print("aten ops with device time by call site, per step: %.1f us, %.1f launches" % (sum(v[1] for _, v in out) / 5, sum(v[0] for _, v in out) / 5))
    for (k, site), (c, t) in out[:60]:
        print("%-28s calls/step %5.1f  us/step %7.1f   %s" % (k, c / 5, t / 5, site))
    sys.exit(0)
rows = [(e.key, e.count, e.self_device_time_total) for e in prof.key_averages() if e.self_device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print("aten ops with device time, 5 steps: total %.1f us per step" % (tot / 5))
for k, c, t in rows[:40]:
    print("%-40s calls/step %5.1f  us/step %8.1f" % (k, c / 5, t / 5))
