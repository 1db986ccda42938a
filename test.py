#!/usr/bin/env python3
"""Evaluation driver -- counterpart of the reference's test.py (`python test.py --opt <yml>`): loads
`experiments/<index>/model.pth`, renders every test view in `test.max_height x max_width` chunks
through PAPR.evaluate on the HIP path, reports PSNR and SSIM (test.py:107-108; papr_amd/metrics.py) and writes PNGs.
LPIPS needs pretrained weights that are not available offline and is not computed."""
import argparse
import os

import numpy as np
import torch

from papr_amd import get_model, load_config
from papr_amd.config import as_node, parse_overrides
from papr_amd.dataset import get_dataset
from papr_amd.metrics import ssim
from train import psnr, render_full


def main():
    ap = argparse.ArgumentParser(description="PAPR")
    ap.add_argument("--opt", type=str, default="")
    ap.add_argument("--max-views", type=int, default=-1)
    ap.add_argument("--save", action="store_true")
    ap.add_argument("--set", nargs="*", default=[], help="extra overrides, e.g. use_amp=false test.max_height=200")
    cli = ap.parse_args()
    cfg = load_config(cli.opt, overrides=parse_overrides(cli.set))
    args = as_node(cfg)
    dev = torch.device("cuda")
    log_dir = os.path.join(args.save_dir, args.index)
    model = get_model(args, dev).to(dev)
    load = os.path.join(args.save_dir, args.test.load_path) if args.test.load_path else log_dir
    step = model.load(load)
    print("loaded step", step, "points", model.points.shape[0])
    results = {}
    for ds in cfg["test"]["datasets"]:
        dcfg = dict(cfg["dataset"]); dcfg.update(ds)
        data = get_dataset(dcfg, ds["mode"], dev, seed=args.seed)
        n = len(data) if cli.max_views < 0 else min(len(data), cli.max_views)
        vals, ssims = [], []
        for i in range(n):
            img, rayd, rayo, c2w = data.full_view(i)
            rgb = render_full(model, rayo, rayd, c2w, args.test.max_height, args.test.max_width)
            vals.append(psnr(rgb, img))
            ssims.append(ssim(rgb[0].cpu().numpy(), img[0].cpu().numpy()))
            if cli.save:
                from PIL import Image
                out = os.path.join(log_dir, "test", ds["name"]); os.makedirs(out, exist_ok=True)
                Image.fromarray((rgb[0].cpu().numpy() * 255).astype(np.uint8)).save(os.path.join(out, "%03d.png" % i))
        results[ds["name"]] = float(np.mean(vals))
        print("testset", ds["name"], "views", n, "avg psnr", results[ds["name"]], "avg ssim", float(np.mean(ssims)))
    return results


if __name__ == "__main__":
    main()
