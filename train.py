#!/usr/bin/env python3
"""Training driver on the HIP render path -- same command line, schedule and checkpoint files as the
reference's train.py (`python train.py --opt configs/nerfsyn/chair.yml [--resume 1]`; add `--set training.losses.lpips=0` when the LPIPS weight
files are not available, see papr_amd/lpips.py -- such runs train MSE-only):
prune / add schedule (train.py:207-250), train_step call order (:155-179), evaluation every
`eval.step` steps (and every 500 below 10,000) with chunked full-image rendering (:29-152), rank-0
logging and checkpoints.  Launch under torch.distributed.run for ray-sharded data parallelism.
Matplotlib dashboards / videos of the reference are not reproduced (diagnostics, out of scope).
"""
import argparse
import bisect
import os
import sys
import time

import numpy as np
import torch

from papr_amd import dist as pdist, get_loss, get_model, load_config
from papr_amd.config import as_node, eval_config, parse_overrides
from papr_amd.dataset import get_dataset, sample_batch


def parse_args():
    ap = argparse.ArgumentParser(description="PAPR")
    ap.add_argument("--opt", type=str, default="", help="Option file path")
    ap.add_argument("--resume", type=int, default=0, help="Resume training")
    ap.add_argument("--steps", type=int, default=-1, help="override training.steps (smoke runs)")
    ap.add_argument("--log-steps", type=str, default="", help="diagnostics: write every step's loss / image index / point count and every prune / add event to this .npz "
                                                             "(reads the loss back every step: not for timing runs)")
    ap.add_argument("--set", nargs="*", default=[], help="extra overrides, e.g. use_amp=false training.losses.lpips=0")
    return ap.parse_args()


def setup_seed(seed):
    import random
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


EVAL_CHUNK_RAYS = 160000          # rays per evaluate() call the drivers aim for on a 288 GB device (the reference's 200 x 200 suits 24 GB)


def eval_chunk(model, N, H, W, max_h, max_w, device):
    """The chunk of an image that one evaluate() call renders.  The configured `max_height x max_width` is the reference's memory
    limit, not part of the result -- evaluate() is chunk-invariant bit for bit (tests/test_hip_model.py) and the render head runs once
    on the whole map --, so on a device with room the drivers take larger chunks: fewer, longer launches (800 x 800 lego view: 84.3 ms
    in 200 x 200 chunks, 79.7 in 400 x 400).  PAPR_EVAL_CHUNK=config keeps the configured size."""
    if os.environ.get("PAPR_EVAL_CHUNK", "auto") == "config" or device.type != "cuda":
        return max_h, max_w
    k = min(model.points.shape[0], int(model.select_k))
    per_ray = 4 * k * 1024                      # bytes: the pair rows of a ray (inputs, two 256-wide ping-pong rows, value row) with room to spare
    free = torch.cuda.mem_get_info(device)[0]
    h, w = min(max_h, H), min(max_w, W)
    while True:                                 # double the smaller side while the chunk stays inside the image, the target and a quarter of the free memory
        nh, nw = (min(2 * h, H), w) if h <= w and h < H else (h, min(2 * w, W))
        if (nh, nw) == (h, w) and h < H:
            nh = min(2 * h, H)
        if (nh, nw) == (h, w) or N * nh * nw > EVAL_CHUNK_RAYS or N * nh * nw * per_ray > free // 4:
            return h, w
        h, w = nh, nw


def render_full(model, rayo, rayd, c2w, max_h, max_w, extras=False):
    """Chunked evaluate() + render head + compositing (reference eval_step / test_step).  extras: also the foreground image, the
    background mask, the attention weights and the selected points (what test.py:120-141 turns into depth / fgrgb / bkgmask PNGs)."""
    args = model.args
    N, H, W, _ = rayd.shape
    max_h, max_w = eval_chunk(model, N, H, W, max_h, max_w, rayd.device)
    topk = min(model.points.shape[0], int(model.select_k))
    C = args.models.attn.embed.value.d_ff_out
    fmap = torch.zeros(N, H, W, 1, C, device=rayd.device)
    attn = torch.zeros(N, H, W, topk + 1, 1, device=rayd.device)
    sel = torch.zeros(1, H, W, topk, 3, device=rayd.device) if extras else None
    with torch.no_grad():
        for h0 in range(0, H, max_h):
            for w0 in range(0, W, max_w):
                f, a = model.evaluate(rayo, rayd[:, h0:h0 + max_h, w0:w0 + max_w].contiguous(), c2w)
                fmap[:, h0:h0 + max_h, w0:w0 + max_w] = f
                attn[:, h0:h0 + max_h, w0:w0 + max_w] = a
                if extras:
                    sel[:, h0:h0 + max_h, w0:w0 + max_w] = model.selected_points[:1]
        if args.models.use_renderer:
            fg = model.renderer(fmap.squeeze(-2).permute(0, 3, 1, 2)).permute(0, 2, 3, 1).unsqueeze(-2).float()
        else:
            fg = fmap
        bkg_attn = attn[..., topk:, :]
        bkg = model.bkg_feats.expand(N, H, W, -1, -1)
        rgb = fg * (1 - bkg_attn) + bkg * bkg_attn if args.models.normalize_topk_attn else fg + bkg * bkg_attn
        out = torch.clamp(model.last_act(rgb.squeeze(-2)), 0, 1)
        if extras:
            return out, fg.squeeze(-2), (bkg * bkg_attn).squeeze(-2), attn, sel
        return out


def psnr(rgb, img):
    return -10.0 * np.log(((rgb - img) ** 2).mean().item()) / np.log(10.0)


def train_step(step, model, batch, loss_fn, args):
    tgt, rayd, rayo, c2w = batch
    model.clear_grad()
    out = model.last_act(model(rayo, rayd, c2w, step))
    loss = loss_fn(out, tgt)
    model.scaler.scale(loss).backward()
    nan_from = int(os.environ.get("PAPR_DEBUG_NANCHECK_FROM", "-1"))
    if 0 <= nan_from <= step and not model.scaler.is_enabled():      # (debugging aid: the first step whose gradients are not finite -- its inputs and the model go to PAPR_DEBUG_DUMP)
        bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        if bad or not torch.isfinite(loss):
            print("step %d: loss %r, non-finite gradients in %d tensors: %s" % (step, float(loss), len(bad), bad[:8]), flush=True)
            if os.environ.get("PAPR_DEBUG_DUMP"):
                torch.save({"step": step, "state": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "batch": [b.detach().cpu() for b in batch],
                            "grads": {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}}, os.environ["PAPR_DEBUG_DUMP"])
            sys.exit(3)
    model.step(step)
    if args.scaler_min_scale > 0 and model.scaler.get_scale() < args.scaler_min_scale:
        model.scaler.update(args.scaler_min_scale)
    else:
        model.scaler.update()
    return loss


def reinit(model, step, fn):
    model.clear_optimizer()
    model.clear_scheduler()
    out = fn()
    model.init_optimizers(step)
    return out


def main():
    sys.stdout.reconfigure(line_buffering=True)      # (a log that goes to a file must end where the run ended: a block-buffered one once hid 4,600 steps in front of a GPU fault)
    cli = parse_args()
    cfg = load_config(cli.opt, overrides=parse_overrides(cli.set))
    if cli.steps > 0:
        cfg["training"]["steps"] = cli.steps
    args, eargs = as_node(cfg), as_node(eval_config(cfg))
    world = pdist.init_from_env("cuda")
    rank = pdist.rank()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    log_dir = os.path.join(args.save_dir, args.index)
    if rank == 0:
        os.makedirs(log_dir, exist_ok=True)
    setup_seed(args.seed)
    model = get_model(args, dev).to(dev)
    pdist.broadcast_module_state(model)
    dataset = get_dataset(cfg["dataset"], "train", dev, seed=args.seed + rank, own_stream=world > 1)
    eval_set = get_dataset(eargs["dataset"], "test", dev, seed=args.seed)
    loss_fn = get_loss(cfg["training"]["losses"]).to(dev)
    start = 0
    eval_psnrs, train_losses, eval_losses = [], [], []
    if cli.resume > 0:                          # same files as the reference (train.py:313-322): either driver resumes the other's directory
        start = model.load(log_dir)
        hist = {}
        for name in ("train_losses", "eval_losses", "eval_psnrs"):
            path = os.path.join(log_dir, name + ".pth")
            hist[name] = torch.load(path).tolist() if os.path.exists(path) else []
        train_losses, eval_losses, eval_psnrs = hist["train_losses"], hist["eval_losses"], hist["eval_psnrs"]
        print("!!!!! Resume from step %s" % start)
    elif args.load_path:                        # start from another experiment's weights (train.py:324-332)
        load_dir = os.path.join(args.save_dir, args.load_path)
        try:
            loaded = model.load(load_dir)
        except Exception:
            for loaded, sd in torch.load(os.path.join(load_dir, "model.pth"), map_location="cpu").items():
                model.load_my_state_dict(sd)
        print("!!!!! Loaded model from %s at step %s" % (args.load_path, loaded))
    T = args.training
    step, pruned, t0, run_loss = start, False, time.time(), 0.0
    print("Start step:", start, "Total steps:", T.steps)
    slog = {"loss": [], "P": [], "events": [], "img": [], "tgt_sum": []} if (cli.log_steps and rank == 0) else None
    while step < T.steps:
        # (the reference's `for batch in trainloader` has the step's batch in hand BEFORE the prune / add block runs, train.py:205-250: the crop is
        # drawn from the global numpy stream ahead of add_points' draws)
        batch = sample_batch(dataset, cfg["dataset"]["batch_size"])
        if T.prune_steps > 0 and T.prune_start <= step < T.prune_stop and step % T.prune_steps == 0:
            thr = T.prune_thresh_list[bisect.bisect_left(T.prune_steps_list, step)] if len(T.prune_steps_list) > 0 else T.prune_thresh
            n = reinit(model, step, lambda: model.prune_points(thr))
            pruned = True
            print("Step %d: Pruned %d points" % (step, n))
            if slog is not None:
                slog["events"].append((step, 0, int(n), int(model.points.shape[0])))
        add_now = None
        if pruned and len(T.add_steps_list) > 0:
            if step in T.add_steps_list:
                add_now = T.add_num_list[T.add_steps_list.index(step)]
        elif pruned and T.add_steps > 0 and step % T.add_steps == 0 and T.add_start <= step < T.add_stop:
            add_now = T.add_num
        if add_now is not None:
            capped = min(add_now, args.max_num_pts - model.points.shape[0]) if args.max_num_pts > 0 else add_now
            if capped > 0:      # the reference passes the un-capped count on (train.py:247)
                n = reinit(model, step, lambda: model.add_points(add_now if len(T.add_steps_list) == 0 else capped))
                model.added_points = True
                print("Step %d: Added %d points" % (step, n))
                if slog is not None:
                    slog["events"].append((step, 1, int(n), int(model.points.shape[0])))
        if slog is not None:
            slog["P"].append(int(model.points.shape[0]))
        if 0 <= int(os.environ.get("PAPR_DEBUG_SYNC_FROM", "-1")) <= step:      # (debugging aid: papr_amd/hip.py: DEBUG_SYNC)
            from papr_amd import hip as _hip
            _hip.DEBUG_SYNC = True
            if os.environ.get("PAPR_DEBUG_DUMP"):       # the step's inputs, overwritten every step: what the faulting step was given
                torch.save({"step": step, "points": model.points.detach().cpu(), "influ": model.points_influ_scores.detach().cpu(),
                            "batch": [b.detach().cpu() for b in batch]}, os.environ["PAPR_DEBUG_DUMP"])
        loss = train_step(step, model, batch, loss_fn, args)
        step += 1
        if slog is not None:
            slog["loss"].append(loss.item())
            slog["img"].append(getattr(dataset, "last_indices", [-1])[0])
            slog["tgt_sum"].append(float(batch[0].double().sum()))
        if step % 200 == 0 and rank == 0:
            l = loss.item()
            print("Train step:", step, "loss:", l, "attn_lr:", model.attn_lr, "pts_lr:", model.pts_lr, "scale:",
                  model.scaler.get_scale(), "points:", model.points.shape[0], f"time: {time.time() - t0:.2f}s")
            t0 = time.time()
        if (step % args.eval.step == 0) or (step % 500 == 0 and step < 10000) or step == T.steps:
            if rank == 0:
                img, rayd, rayo, c2w = eval_set.full_view(args.eval.img_idx % len(eval_set))
                rgb = render_full(model, rayo, rayd, c2w, args.eval.max_height, args.eval.max_width)
                eval_psnrs.append(psnr(rgb, img))
                eval_losses.append(loss_fn(rgb, img).item())
                train_losses.append(loss.item())
                print("Eval step:", step, "train_loss:", train_losses[-1], "eval_psnr:", eval_psnrs[-1])
                model.save(step, log_dir)
                torch.save(torch.tensor(train_losses), os.path.join(log_dir, "train_losses.pth"))
                torch.save(torch.tensor(eval_losses), os.path.join(log_dir, "eval_losses.pth"))
                torch.save(torch.tensor(eval_psnrs), os.path.join(log_dir, "eval_psnrs.pth"))
                if step % 50000 == 0:
                    torch.save(model.state_dict(), os.path.join(log_dir, "model_%d.pth" % step))
            if pdist.active():
                torch.distributed.barrier()
    if slog is not None:
        np.savez(cli.log_steps, loss=np.array(slog["loss"]), img=np.array(slog["img"], dtype=np.int32), tgt_sum=np.array(slog["tgt_sum"]), P=np.array(slog["P"], dtype=np.int32), events=np.array(slog["events"], dtype=np.int32).reshape(-1, 4),
                 eval_psnrs=np.array(eval_psnrs), eval_losses=np.array(eval_losses), points_final=model.points.detach().cpu().numpy(),
                 influ_final=model.points_influ_scores.detach().cpu().numpy(), attn_lr=np.array(model.attn_lr), pts_lr=np.array(model.pts_lr))
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return eval_psnrs


if __name__ == "__main__":
    main()
