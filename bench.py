#!/usr/bin/env python3
"""Headline benchmark: PAPR training throughput (rays/s) on nerf_synthetic/chair-shaped work.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = the reference's train_step (train.py:155-179): clear grads, forward of one 160x160 patch
(R = 25,600 rays against P = 10,000 points, k = 20), MSE loss, backward, the five Adam groups, the
schedulers.  Rays and targets are generated on the device before the timed region (synthetic scene,
see papr_amd/data.py).  With N > 1 every rank renders its own patch (weak scaling) and gradients are
averaged with one RCCL all-reduce inside PAPR.step().

The JSON line also carries
  roofline      the dominant kernel (default mode: mlp_chain4_kernel, the fused embedding-MLP runs, forward and
                data-gradient), timed live with HIP events on the launch stream during the timed steps; `frac` counts the
                three f16 MFMA products of every fp32 product as work, `frac_algorithmic` only the fp32 flops
  roofline_knn  the ray -> k-nearest-points kernel against its logical HBM byte count (north_star)
  throughput_mode_h1   the same run in the reduced-precision mode of the fused MLP runs (a child process; never `value`)
  cpu_baseline  the CPU oracle's train step (torch fp32, same math) on this host's cores, on a
                bounded sample (1,024 rays against the same 10,000-point cloud)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TF = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 FLOP/clk/CU x 256 CU x 2.4 GHz
HBM_PEAK_GBS = 8000.0         # spec; 6.29 TB/s measured copy ceiling
F16_MFMA_PEAK_TF = 2500.0     # MI355X_MICROARCH.md: dense f16/bf16 MFMA (v_mfma_f32_32x32x16_f16, 32 cycles per SIMD)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=10000, help="point-cloud size (chair.yml: 10000 at init, <=30000 late)")
    ap.add_argument("--scene", default="nerfsyn/chair.yml")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm-mode", default="h3", choices=["f32", "fwd", "dgrad", "layers", "h3", "h1", "h3_f16rows"],
                    help="h3 (default): split-f16 MFMA everywhere, consecutive layers fused into one launch; layers: the same "
                         "arithmetic, one launch per layer; dgrad: forward + data-gradient only; fwd: forward only; f32: fp32 MFMA everywhere; "
                         "h1: h3 with one f16 product per fp32 product in the fused runs (reduced precision, reported as throughput_mode_h1)")
    ap.add_argument("--h3-rows", default="f16", choices=["f16", "f32"],
                    help="gemm mode h3: what a fused run keeps for its weight gradients -- f16 (default since round 6, PAPR_MLP_H3_F16ROWS: forward and data-gradient "
                         "arithmetic unchanged, every golden bar unchanged) or f32 (PAPR_MLP_H3: rounds 1-5; reported as parity_fp32_rows in the default line)")
    ap.add_argument("--no-amp-line", action="store_true", help="skip the second measurement (throughput_mode_h1: a child process in --gemm-mode h1)")
    ap.add_argument("--cpu-rays", type=int, default=32, help="edge of the CPU-baseline patch (32 -> 1,024 rays)")
    ap.add_argument("--cpu-steps", type=int, default=8)
    ap.add_argument("--psnr-steps", type=int, default=3000,
                    help="N = 1 only: after the timed region, train.py runs this many steps of the same scene file from seed 1 on the procedural scene "
                         "(a child process) and its last evaluation PSNR goes into the JSON line as psnr_after_steps (BASELINE's metric is rays/s + test PSNR); 0 = skip")
    ap.add_argument("--profile-steps", type=int, default=-1,
                    help="steps of the second, instrumented pass behind the timed region (HIP events around every library launch: the roofline records); "
                         "-1 = as many as --steps, 0 = none (no roofline objects)")
    ap.add_argument("--no-shipped-line", action="store_true", help="skip as_shipped_amp (a child process with --amp: the scene file's own use_amp: true)")
    ap.add_argument("--amp", action="store_true",
                    help="run with the scene file's own `use_amp: true` (the reference trains its attention block and U-Net under fp16 autocast then; "
                         "here: GradScaler on, embedding MLPs in the one-product mode, U-Net on the own split-f16 kernels); default is the fp32 "
                         "parity mode the 1e-4 bar is stated for")
    return ap.parse_args()


def bench_config(scene, points, amp=False):
    from papr_amd import load_config
    # fp32 parity mode: no autocast anywhere; LPIPS needs VGG weights that cannot be fetched offline
    return load_config(scene, overrides={"use_amp": bool(amp), "geoms": {"points": {"init_num": points}},
                                         "training": {"losses": {"mse": 1.0, "lpips": 0.0, "lpips_alex": 0.0}}})


def train_step(model, loss_fn, batch, step):
    tgt, rayd, rayo, c2w = batch
    model.clear_grad()
    out = model.last_act(model(rayo, rayd, c2w, step))
    loss = loss_fn(out, tgt)
    model.scaler.scale(loss).backward()
    model.step(step)
    model.scaler.update()
    return loss


def cpu_baseline(cfg, state, edge, steps):
    """Oracle (CPU restatement of the reference) train steps on a bounded sample of the same workload."""
    from oracle import papr_oracle as O
    from papr_amd.data import SyntheticRayData
    import copy
    cfg = copy.deepcopy(cfg)
    cfg["dataset"]["patches"] = {"height": edge, "width": edge, "max_patches": 1}
    torch.set_num_threads(min(os.cpu_count(), 16))   # 16 threads is the fastest setting on the 256-thread EPYC host (scripts/cpu_threads_sweep.py)
    data = SyntheticRayData(cfg["dataset"], n_views=8, seed=3, device="cpu")
    st = O.trainable_state({k: v.detach().cpu() for k, v in state.items()}, cfg)
    opts = O.make_optimizers(st, cfg)
    batches = [data.patch() for _ in range(2)]
    O.train_step(st, opts, cfg, batches[0][2], batches[0][1], batches[0][0])       # warm-up
    t0 = time.perf_counter()
    for i in range(steps):
        tgt, rayd, rayo, _ = batches[i % 2]
        O.train_step(st, opts, cfg, rayo, rayd, tgt)
    dt = time.perf_counter() - t0
    R = edge * edge
    return {"value": R * steps / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port",
            "sample": "%d train steps of %dx%d=%d rays vs P=%d (oracle/papr_oracle.py, torch %s fp32 CPU, Adam, MSE)"
                      % (steps, edge, edge, R, state["points"].shape[0], torch.__version__)}


_LAUNCHER_KEYS = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME",
                  "MASTER_ADDR", "MASTER_PORT", "PAPR_DIST_SINGLE", "PAPR_DIST_BACKEND", "OMP_NUM_THREADS")


def child_env():
    """The environment of a one-process child (child_line, psnr_after_steps): this process's own, WITHOUT what a launcher exported for it -- a child
    that inherited RANK / MASTER_* would form another group on the store and port its parent has just used."""
    return {k: v for k, v in os.environ.items() if k not in _LAUNCHER_KEYS and not k.startswith("TORCHELASTIC_")}


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks the way the driver's own command line does
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...) as a fresh CHILD process -- this process has not touched the GPU
    (torch.cuda.device_count() does not initialise it) and never execs -- relay rank 0's JSON line and the child's exit code."""
    import socket
    import subprocess
    n_vis = torch.cuda.device_count()
    share = share_one_gpu() and n_vis >= 1
    if n_vis < args.gpus and not share:
        print("bench.py: --gpus %d but only %d GPU(s) visible here: refusing to print a line for fewer ranks than asked for" % (args.gpus, n_vis), file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: no launcher environment, starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    env = child_env()
    if share:
        env["PAPR_DIST_BACKEND"] = "gloo"              # (RCCL refuses two ranks per device)
        print("bench.py: PAPR_BENCH_SHARE_GPU=1 -- %d ranks on %d GPU(s) over gloo: a TEST of the N > 1 code path, not a measurement" % (args.gpus, n_vis), file=sys.stderr)
    r = subprocess.run(cmd, env=env, cwd=ROOT)
    return r.returncode


def share_one_gpu():
    """PAPR_BENCH_SHARE_GPU=1 (tests/test_hip_rccl.py only): the ranks of `--gpus N` may share the visible GPU(s), process group gloo -- so that the
    world > 1 arithmetic of this file (value = world x R x steps / max-over-ranks time, ranks_seen, the relay of rank 0's line and of the exit code
    through self_launch) runs on a 1-GPU box before the first real multi-GPU run.  The line it prints says so (`test_override`)."""
    return os.environ.get("PAPR_BENCH_SHARE_GPU", "0") == "1"


def psnr_after_steps(args):
    """Quality leg of BASELINE's metric ("train rays/sec + test PSNR"): train.py (the reference's driver protocol: patch sampling, eval
    every 500 steps below 10,000, full-image chunked render, test.py:107 PSNR) for --psnr-steps steps, fixed seed, same kernels."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [sys.executable, os.path.join(ROOT, "train.py"), "--opt", os.path.join(ROOT, "configs", args.scene), "--steps", str(args.psnr_steps),
               "--set", "use_amp=%s" % ("true" if args.amp else "false"), "training.losses.lpips=0", "seed=1", "index=bench_psnr", "save_dir=%s" % tmp,
               "geoms.points.init_num=%d" % args.points]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, cwd=ROOT, env=child_env())
        wall = time.perf_counter() - t0
    evals = [l for l in r.stdout.splitlines() if l.startswith("Eval step:")]
    if r.returncode != 0 or not evals:
        return {"error": (r.stderr or r.stdout)[-300:]}
    last = evals[-1].split()
    return {"steps": int(last[2]), "eval_psnr_db": float(last[-1]), "first_eval_psnr_db": float(evals[0].split()[-1]), "train_loss": float(last[4]),
            "wall_s_incl_evals_and_startup": wall, "seed": 1,
            "scene": "procedural chair-like solid on white (papr_amd/data.py; no dataset offline), 100 train views, eval view img_idx 50 at 800x800, "
                     "train.py protocol (MSE-only), PSNR = test.py:107 formula",
            "note": "seed spread of this quantity over long runs: profiles/r03_seed_study/summary.txt"}


def main():
    args = parse()
    os.environ["PAPR_GEMM_MODE"] = args.gemm_mode       # read by papr_amd/ops.py (mlp_mode) at every call: the `mode` argument of papr_mlp_fwd / _bwd
    os.environ["PAPR_H3_ROWS"] = args.h3_rows
    from papr_amd import dist as pdist, get_model, get_loss, hip
    from papr_amd.data import SyntheticRayData
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus %d" % args.gpus)
    if args.gpus > 1 and not pdist.launched():
        sys.exit(self_launch(args))                     # (before anything here touches the GPU)
    if pdist.launched() and int(os.environ["WORLD_SIZE"]) != args.gpus:
        # a line that says dp<WORLD_SIZE> under a command that asked for --gpus N would be read as an N-GPU number
        if int(os.environ.get("RANK", "0")) == 0:
            print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks: the two must agree" % (args.gpus, os.environ["WORLD_SIZE"]), file=sys.stderr)
        sys.exit(2)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if share_one_gpu() and pdist.launched() and torch.cuda.device_count() >= 1:
        if os.environ.get("PAPR_DIST_BACKEND") != "gloo":
            raise SystemExit("bench.py: PAPR_BENCH_SHARE_GPU=1 needs PAPR_DIST_BACKEND=gloo (RCCL refuses two ranks per device)")
        local = local % torch.cuda.device_count()
    if torch.cuda.device_count() <= local:
        print("bench.py: rank with LOCAL_RANK %d but %d GPU(s) visible (the render path has no CPU fallback; one GPU per rank)" % (local, torch.cuda.device_count()), file=sys.stderr)
        sys.exit(2)
    world = pdist.init_from_env("cuda")
    rank = pdist.rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device (the render path has no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    assert world == args.gpus, (world, args.gpus)

    cfg = bench_config(args.scene, args.points, args.amp)
    torch.manual_seed(cfg["seed"])
    import numpy as np
    np.random.seed(cfg["seed"])
    devnull = open(os.devnull, "w")
    stdout, sys.stdout = sys.stdout, devnull           # the model prints its LR banner like the reference
    model = get_model(cfg, device="cpu")
    sys.stdout = stdout
    with torch.no_grad():                               # untrained influence is exactly 0: give the scores work to do
        model.points_influ_scores.uniform_(0.0, 1.0, generator=torch.Generator().manual_seed(5))
    init_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    model.clear_optimizer(); model.clear_scheduler()    # optimizers over the device tensors, as `get_model(args, "cuda")` creates them
    sys.stdout = devnull
    model.init_optimizers(0)
    sys.stdout = stdout
    pdist.broadcast_module_state(model)
    loss_fn = get_loss(cfg["training"]["losses"]).to(dev)

    data = SyntheticRayData(cfg["dataset"], n_views=100, seed=100 + rank, device=dev)
    pool = [data.patch() for _ in range(8)]             # resident in HBM before the timed region
    N, H, W, _ = pool[0][1].shape
    R = N * H * W
    k = int(cfg["geoms"]["points"]["select_k"])
    P = model.points.shape[0]

    def barrier():
        if torch.distributed.is_initialized():
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        train_step(model, loss_fn, pool[i % len(pool)], i)
    # ---- the timed region: exactly `steps` steps of the product as it ships, no instrumentation ----
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = train_step(model, loss_fn, pool[i % len(pool)], args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    final_loss = float(loss.detach())
    # ---- a second pass for the roofline records: the library brackets every launch with a HIP event pair on the launch stream ----
    prof_steps = args.steps if args.profile_steps < 0 else args.profile_steps
    recs, dt_prof = [], 0.0
    if prof_steps > 0:
        hip.profile_enable(True)
        barrier()
        t1 = time.perf_counter()
        for i in range(prof_steps):
            train_step(model, loss_fn, pool[i % len(pool)], args.warmup + args.steps + i)
        barrier()
        dt_prof = time.perf_counter() - t1
        hip.profile_enable(False)
        recs = hip.profile_collect()
    ranks_seen = 1
    if torch.distributed.is_initialized():
        cdev = dev if torch.distributed.get_backend() == "nccl" else "cpu"      # (gloo -- PAPR_BENCH_SHARE_GPU -- reduces host tensors)
        t = torch.tensor([dt], device=cdev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
        one = torch.ones(1, device=cdev, dtype=torch.float64)             # every rank that ran the timed steps adds itself
        torch.distributed.all_reduce(one, op=torch.distributed.ReduceOp.SUM)
        ranks_seen = int(round(float(one)))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if ranks_seen != args.gpus:
        if rank == 0:
            print("bench.py: %d ranks took part, --gpus %d" % (ranks_seen, args.gpus), file=sys.stderr)
        sys.exit(3)
    if rank != 0:
        return

    # ---- roofline of the dominant kernel, from the HIP-event records of the timed steps ---------
    plan = model.plan
    true_k = {plan.key.ld_in: plan.key_w, plan.val.ld_in: plan.val_w, plan.qry.ld_in: plan.qry_w}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    traffic_all = json.load(open(tpath)) if os.path.exists(tpath) else {}
    # (the counter passes were run in the parity mode; the one-product modes move f16 rows: traffic null there)
    traffic_db = traffic_all if (args.gemm_mode != "h1" and not args.amp and args.h3_rows == traffic_all.get("h3_rows", "f32")) else {}
    dt_prof = max(dt_prof, 1e-9)

    def mfma_line(kernel, ids, key):
        rs = [r for r in recs if r[0] in ids]
        ms = sum(r[4] for r in rs)
        fl = sum(2.0 * M * min(Nn, 256) * true_k.get(K, K) for _, M, Nn, K, *_ in rs)
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        return ms, {"kernel": kernel, "bound": "mfma", "achieved": ach, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": ach / FP32_MFMA_PEAK_TF, "traffic": traffic_db.get(key), "launches": len(rs),
                    "avg_launch_ms": ms / max(len(rs), 1), "algorithmic_gflop_per_launch": fl / max(len(rs), 1) / 1e9,
                    "share_of_step_time": ms / (dt_prof * 1e3)}

    def h3_line():
        # split-f16 GEMM: 3 f16 MFMAs per fp32 product make the layer HBM-bound.  Algorithmic bytes per launch:
        # A read once (M x K fp32), C written once (M x N), the weight once, plus the M x N activation-derivative
        # mask of a data-gradient launch.
        rs = [r for r in recs if r[0] in (6, 7)]
        ms = sum(r[4] for r in rs)
        by = sum(4.0 * (M * (true_k.get(K, K) + Nn + (Nn if kid == 7 else 0)) + Nn * K) for kid, M, Nn, K, *_ in rs)
        fl = sum(2.0 * M * Nn * true_k.get(K, K) for _, M, Nn, K, *_ in rs)
        ach = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return ms, {"kernel": "gemm_nt_h3_kernel<128,256,2,2,4> (embedding-MLP forward + data-gradient GEMMs, split-f16 MFMA)",
                    "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic_db.get("gemm_nt_h3_bytes_per_launch"), "launches": len(rs),
                    "avg_launch_ms": ms / max(len(rs), 1), "algorithmic_bytes_per_launch": by / max(len(rs), 1),
                    "fp32_equivalent_tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, "share_of_step_time": ms / (dt_prof * 1e3)}

    def wgrad_h3_line():
        # split-f16 weight gradient: G (M x N) and X (M x K) are read once, the 256 per-CU partial tiles are
        # written once and read once by the reduction
        rs = [r for r in recs if r[0] == 8]
        ms = sum(r[4] for r in rs)
        # (one record per launch = a batch of weight-gradients; the library sums 4 M (N + K) bytes and 2 M N K flops over its jobs)
        by = float(sum(r[5] for r in rs))
        fl = float(sum(r[6] for r in rs))
        ach = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        wg_prods = 1.0 if (args.amp or args.gemm_mode == "h1" or (args.gemm_mode == "h3" and args.h3_rows == "f16")) else 3.0
        return ms, {"kernel": "gemm_tn_tr_kernel / gemm_tn_tr_n32_kernel / gemm_tn_h3_kernel (weight gradients of a fused run, one slice of the rows per CU; with f16 rows the full "
                              "256 x 256 layers by LDS-DMA + transposing LDS reads, first layers and fp32 rows register-staged)",
                    "jobs": int(sum(r[2] for r in rs)),
                    "bound": "hbm" if wg_prods == 1.0 else "hbm+split", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic_db.get("gemm_tn_h3_bytes_per_launch"), "launches": len(rs),
                    "avg_launch_ms": ms / max(len(rs), 1), "algorithmic_bytes_per_launch": by / max(len(rs), 1),
                    "fp32_equivalent_tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, "share_of_step_time": ms / (dt_prof * 1e3),
                    # how busy the matrix pipe is beside it: f16 products issued per fp32 product (1 with f16 rows or under use_amp, 3 with fp32 rows) over the dense f16 peak
                    "products_per_fp32_product": wg_prods, "frac_issued": (wg_prods * fl / (ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TF) if ms > 0 else 0.0,
                    "note": "bound: with f16 rows (the default since round 6, and under use_amp) the kernels stream 2 B per operand element: the batches of full layers "
                            "5.2-5.4 TB/s (gemm_tn_tr_kernel), the launch average lower by the first layers' register-staged launches; the same LDS-DMA request stream alone "
                            "reaches 6.1 TB/s, a plain-load read of the pattern 6.2 (scripts/probes/dma_stream.hip, read_patterns.hip; DESIGN.md Appendix C6).  With fp32 rows "
                            "(--h3-rows f32) the in-kernel hi / lo split and the register transpose bind it beside the loads (850 of 980 us remain with the row loads "
                            "ablated, DESIGN.md Appendix B): `bound` says hbm+split there"
                    }

    def chain_line():
        # fused layer runs (chain4.hip): every fp32 product is three f16 MFMA products, and only the run's input, the
        # saved activations / gradient rows and the masks cross HBM.  The matrix pipe is the nearer roof.
        rs = [r for r in recs if r[0] in (9, 10)]
        ms = sum(r[4] for r in rs)
        by = float(sum(r[5] for r in rs))
        fl = float(sum(r[6] for r in rs))
        prods = 1.0 if (args.gemm_mode == "h1" or args.amp) else 3.0        # f16 MFMA products issued per fp32 product
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0          # fp32-equivalent (algorithmic) TFLOP/s
        big = [r for r in rs if r[1] >= 100000]                   # the runs over the R*k pair rows (key / value, forward / data-gradient)
        big_ms = sum(r[4] for r in big)
        return ms, {"kernel": "mlp_chain4_kernel (fused embedding-MLP runs: forward and data-gradient, %s)" % ("one f16 product per fp32 product" if prods == 1.0 else "split-f16 MFMA, three products per fp32 product"),
                    "bound": "mfma", "achieved": tf, "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F16_MFMA_PEAK_TF,
                    "traffic": traffic_db.get("mlp_chain_bytes_per_launch"), "launches": len(rs),
                    "avg_launch_ms": ms / max(len(rs), 1), "algorithmic_gflop_per_launch": fl / max(len(rs), 1) / 1e9,
                    # (since round 5 the R-row products around w_q / w_k ride on this kernel too: two short launches per step that pull the plain
                    # average down -- the figure comparable with earlier rounds is the one over the four R*k-row runs)
                    "avg_launch_ms_pair_row_runs": big_ms / max(len(big), 1), "launches_pair_row_runs": len(big),
                    # the six runs per step that rounds 1-4 averaged (key / value / query MLP, forward and data-gradient): every launch but the two-layer w_q / w_k run
                    "avg_launch_ms_mlp_runs": sum(r[4] for r in rs if r[2] != 2) / max(len([r for r in rs if r[2] != 2]), 1),
                    "frac_pair_row_runs": (sum(float(r[6]) for r in big) / (big_ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TF) if big_ms > 0 else None,
                    "note": "achieved / frac = ALGORITHMIC flops (2 M N K of every layer, true input widths 117 / 142 / 39) over the launch time, against the dense f16 MFMA peak; "
                            "frac_issued counts the f16 MFMA products the kernel issues per fp32 product (parity mode: hi.hi + hi.lo + lo.hi = 3) = matrix-pipe utilisation",
                    "frac_issued": prods * tf / F16_MFMA_PEAK_TF, "issued_tflops": prods * tf, "products_per_fp32_product": prods,
                    "algorithmic_bytes_per_launch": by / max(len(rs), 1),
                    "hbm_gbs": by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0, "hbm_frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else 0.0,
                    "share_of_step_time": ms / (dt_prof * 1e3)}

    def conv_line():
        # 3x3 layers of the U-Net head (conv.hip): forward / data-gradient (11) and weight-gradient (12) launches incl. their
        # tensor-maximum, weight-split and reduction kernels; three f16 MFMA products per fp32 product
        rs = [r for r in recs if r[0] in (11, 12)]
        ms = sum(r[4] for r in rs)
        if ms <= 0:
            return None
        fl = float(sum(r[6] for r in rs))
        return {"kernel": "conv3x3_h3_kernel + conv3x3_wgrad_h3_kernel (U-Net 3x3 layers, split-f16 implicit GEMM)", "bound": "mfma",
                "achieved": 3.0 * fl / (ms * 1e-3) / 1e12, "peak": F16_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": 3.0 * fl / (ms * 1e-3) / 1e12 / F16_MFMA_PEAK_TF, "traffic": None, "launches": len(rs),
                "avg_launch_ms": ms / len(rs), "fp32_equivalent_tflops": fl / (ms * 1e-3) / 1e12, "share_of_step_time": ms / (dt_prof * 1e3)}

    nt_ms, nt_line = mfma_line("gemm_nt_kernel<128,256,2,2> (embedding-MLP forward + data-gradient GEMMs, fp32 MFMA)", (0,),
                               "gemm_nt_128x256_bytes_per_launch")
    tn_ms, tn_line = mfma_line("gemm_tn_kernel (weight gradients, split over M, fp32 MFMA)", (4,), "gemm_tn_bytes_per_launch")
    h3_ms, h3 = h3_line()
    wg_ms, wg = wgrad_h3_line()
    ch_ms, ch = chain_line()
    dominant = max(((nt_ms, nt_line), (tn_ms, tn_line), (h3_ms, h3), (wg_ms, wg), (ch_ms, ch)), key=lambda t: t[0])[1]
    def knn_line():
        # north_star asks for "the fraction of the kNN HBM roofline".  The binned form reads ~8 % of the cloud per ray and the cloud (120 KB at
        # P = 10,000) is cache-resident, so the HBM roofline does not bind this kernel: the line carries the PHYSICAL fetch rate from the counter
        # pass and the instruction count that does bind it, and the logical rate (what an every-point search would have to move) without a fraction.
        knn = [r for r in recs if r[0] == 5]
        knn_ms = sum(r[4] for r in knn) / max(len(knn), 1)
        knn_bytes = R * (12.0 * P + 12 + 4 * k)
        fetch_kb = traffic_all.get("ray_knn_fetch_kb_raw_P%d" % P)
        insts = traffic_all.get("ray_knn_wave_insts_per_launch_P%d" % P)
        phys = 2.0 * 1024.0 * fetch_kb if fetch_kb else None                 # (FETCH_SIZE doubled: the guide's gfx950 correction)
        return {"kernel": "ray_knn_blocks_kernel (P >= 2,048: binned cloud, bounding spheres; the three binning kernels, ~18 us, are not in avg_launch_ms)",
                "bound": "issue", "unit": "GB/s", "peak": HBM_PEAK_GBS, "avg_launch_ms": knn_ms,
                "achieved": phys / (knn_ms * 1e-3) / 1e9 if phys and knn_ms > 0 else None,
                "frac": phys / (knn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if phys and knn_ms > 0 else None,
                "physical_fetch_bytes_per_launch": phys,
                "physical_fetch_gbs": phys / (knn_ms * 1e-3) / 1e9 if phys and knn_ms > 0 else None,
                "wave_insts_per_ray": insts / R if insts else None,
                "logical_bytes_per_launch": knn_bytes, "logical_gbs": knn_bytes / (knn_ms * 1e-3) / 1e9 if knn_ms > 0 else 0.0,
                "rays_per_s_kernel_alone": R / (knn_ms * 1e-3) if knn_ms > 0 else 0.0,
                "note": "achieved / frac = PHYSICAL HBM-side fetch (rocprofv3 FETCH_SIZE x 2 of the counter pass, profiles/traffic.json) over the live launch time: far "
                        "below the HBM roof because the cloud is cache-resident and the walk touches ~19 of 247 blocks per ray; the kernel is bound by "
                        "instruction issue (wave_insts_per_ray = SQ_INSTS_VALU + SALU + LDS + VMEM of the counter pass / rays).  logical_gbs = (12 P + 12 + 4 k) "
                        "bytes per ray (SURVEY section 8d: what an every-point search reads) over the same time; it exceeds the HBM peak because those bytes are never moved"}

    out = {
        "metric": "train rays/sec, nerf_synthetic/%s (PAPR), fp32 in/out, GEMM mode '%s'" % (os.path.splitext(os.path.basename(args.scene))[0], args.gemm_mode),
        "value": world * R * args.steps / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ranks_seen": ranks_seen,
        **({"test_override": "PAPR_BENCH_SHARE_GPU=1: %d ranks shared %d GPU(s) over gloo -- a test of the N > 1 code path, NOT a measurement" % (world, torch.cuda.device_count())}
           if share_one_gpu() and world > 1 else {}),
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        # north_star's literal figure: "train rays/sec ... as achieved fraction of the kNN HBM roofline" -- LOGICAL bytes (what an every-point search
        # would read per ray, SURVEY section 8d: 12 P + 12 + 4 k) times the END-TO-END train rays/s of the whole job, over N x the HBM peak.  The
        # kernel's own physical traffic is in roofline_knn (the cloud is cache-resident, the binned search reads ~8 % of it)
        "e2e_frac_of_knn_hbm_roofline": {"logical": True, "bytes_per_ray": 12.0 * P + 12 + 4 * k,
                                         "frac": (world * R * args.steps / dt) * (12.0 * P + 12 + 4 * k) / (world * HBM_PEAK_GBS * 1e9),
                                         "note": "value x (12 P + 12 + 4 k) B / (n_gpus x 8 TB/s): end-to-end train throughput priced in the kNN stage's logical HBM bytes; "
                                                 "the step is bound by the embedding MLPs' matrix work (roofline), not by this stream"},
        "dtype": ("f32" if args.gemm_mode == "f32" else "f32 (wide GEMMs: operands split into f16 hi+lo, 3 MFMA, fp32 accumulate"
                  + ("; the rows a fused run keeps for its weight gradients are f16 -- each row's hi plane, one product per weight-gradient term, fp32 accumulate: "
                     "forward, data gradients and rgb parity untouched, gradient bars of the goldens unchanged, tests/test_hip_chain_variants.py" if args.gemm_mode == "h3" and args.h3_rows == "f16" and not args.amp else "") + ")")
                 + ("; use_amp: embedding MLPs one f16 product per fp32 product (PAPR_MLP_H1), GradScaler on, U-Net on the own split-f16 kernels" if args.amp else ""), "data": "synthetic",
        "config": {"workload": "configs/" + args.scene + ": P=%d points, one %dx%d patch (R=%d rays) per rank per step, k=%d, "
                               "U-Net head, MSE loss (LPIPS weight 0: VGG weights unavailable offline), use_amp=%s; "
                               "points_influ_scores drawn U(0,1) (seed 5) instead of the untrained all-zero init so that the attention scores are not all zero"
                               % (P, H, W, R, k, "true (GradScaler, one-product embedding MLPs; the U-Net stays on the own kernels)" if args.amp else "false"),
                   "global_batch_rays": world * R, "parallelism": "dp%d" % world, "gemm_mode": args.gemm_mode, "final_loss": final_loss},
        "roofline": dominant if recs else None,
        "roofline_gemm_nt_fp32": nt_line if nt_ms > 0 and dominant is not nt_line else None,
        "roofline_gemm_nt_h3": h3 if h3_ms > 0 and dominant is not h3 else None,
        "roofline_wgrad": tn_line if tn_ms > 0 and dominant is not tn_line else None,
        "roofline_wgrad_h3": wg if wg_ms > 0 and dominant is not wg else None,
        "roofline_mlp_chain": ch if ch_ms > 0 and dominant is not ch else None,
        "roofline_conv3x3": conv_line(),
        "roofline_knn": knn_line(),
    }
    if os.environ.get("PAPR_BENCH_LAUNCHES"):           # per-shape launch table of the timed steps, on stderr
        tab = {}
        for r in recs:
            c = tab.setdefault(tuple(r[:4]), [0, 0.0])
            c[0] += 1; c[1] += r[4]
        for key in sorted(tab, key=lambda t: -tab[t][1]):
            n, ms = tab[key]
            print("kernel %2d  M=%-8d N=%-5d K=%-5d  %4d launches  %.3f ms each  %.3f ms/step" % (*key, n, ms / n, ms / max(prof_steps, 1)), file=sys.stderr)
    if args.gemm_mode == "h1":
        out["dtype"] = "f32 rows, fused MLP runs multiply one f16 product per fp32 product (fp32 accumulate): the reduced-precision throughput mode"
    def child_line(extra):
        """A second measurement of this script in a fresh process (the mode is fixed when the first model is built); the headline line never depends on it."""
        import subprocess
        cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-amp-line", "--no-shipped-line", "--psnr-steps", "0", "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--scene", args.scene, "--points", str(args.points)] + extra
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=child_env())
            return json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:
            return {"error": "%s: %s" % (type(e).__name__, str(e)[-300:])}

    def chain_of(j):
        return j.get("roofline") if "mlp_chain" in (j.get("roofline") or {}).get("kernel", "") else j.get("roofline_mlp_chain")

    main_line = world == 1 and args.gemm_mode == "h3" and not args.amp and args.h3_rows == "f16"
    if main_line and not args.no_shipped_line:
        j = child_line(["--h3-rows", "f32"])
        out["parity_fp32_rows"] = j if "error" in j else {
            "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "final_loss": j["config"]["final_loss"], "roofline_mlp_chain": chain_of(j),
            "note": "PAPR_H3_ROWS=f32 (mode PAPR_MLP_H3): fp32 rows between a fused run and its weight gradients, three f16 products per weight-gradient term -- the "
                    "default of rounds 1-5.  The headline keeps f16 rows there (PAPR_MLP_H3_F16ROWS): same forward / data-gradient bits, same golden bars "
                    "(G7 gradients, 3-step trajectory, G15 360 steps), 21,500-step PSNR 24.99 / 28.24 / 28.58 against 24.86 / 27.89 / 28.86 dB over seeds 1-3 "
                    "(profiles/r06_seed_study/)"}
    if main_line and not args.no_shipped_line:
        # BASELINE configs[1] verbatim: configs/nerfsyn/chair.yml as shipped has use_amp: true
        j = child_line(["--amp"])
        out["as_shipped_amp"] = j if "error" in j else {
            "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "final_loss": j["config"]["final_loss"], "dtype": j["dtype"],
            "workload": j["config"]["workload"], "roofline_mlp_chain": chain_of(j),
            "note": "the scene file's own `use_amp: true` (the reference then runs its attention block and U-Net under fp16 autocast, models/attn.py:248, "
                    "models/unet.py:212): GradScaler on, embedding MLPs one f16 product per fp32 product, U-Net on the own split-f16 kernels.  Pinned to the "
                    "reference's OWN AMP output (G17: tests/golden/g17_amp_*.npz from make_golden.py --amp, the reference under CPU fp16 autocast; bar = the "
                    "reference's AMP-vs-fp32 distance, tests/test_hip_amp_golden.py); NOT the headline `value`, which is the fp32 parity mode the 1e-4 bar is stated for"}
    if main_line and not args.no_amp_line:
        j = child_line(["--gemm-mode", "h1"])
        out["throughput_mode_h1"] = j if "error" in j else {
            "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "final_loss": j["config"]["final_loss"],
            "note": "PAPR_GEMM_MODE=h1: one f16 product per fp32 product in the fused embedding-MLP runs, GradScaler off; tolerance in tests/test_hip_h1.py; NOT the headline `value`",
            "roofline_mlp_chain": chain_of(j)}
    if main_line and args.psnr_steps > 0:
        try:
            out["psnr_after_steps"] = psnr_after_steps(args)
        except Exception as e:
            out["psnr_after_steps"] = {"error": "%s: %s" % (type(e).__name__, str(e)[-300:])}
    if not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(cfg, init_state, args.cpu_rays, args.cpu_steps)
        except Exception as e:
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, str(e)[-300:])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
