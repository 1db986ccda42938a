import os, sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/golden")
from oracle import papr_oracle as O
from oracle.state import empty_state
from formula import formula_fill, synth_rays, uniform_points
from papr_amd import load_config
cfg = load_config("nerfsyn/chair.yml", overrides={"use_amp": False, "training": {"losses": {"mse": 1.0, "lpips": 0.0}}})
st0 = formula_fill(empty_state(cfg, 10000)); st0["points"] = uniform_points(10000, 12.0, 1)
ro, rd, _ = synth_rays(1, 32, 32, seed=1); tgt = torch.rand(1, 32, 32, 3)
print("cpu_count", os.cpu_count())
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    st = O.trainable_state(st0, cfg); opts = O.make_optimizers(st, cfg)
    O.train_step(st, opts, cfg, ro, rd, tgt)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 6 and n < 6:
        O.train_step(st, opts, cfg, ro, rd, tgt); n += 1
    dt = time.perf_counter() - t0
    print("threads", nt, "steps", n, "rays/s", 1024 * n / dt, flush=True)
