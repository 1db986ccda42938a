#!/usr/bin/env python3
"""Evaluation driver -- counterpart of the reference's test.py (`python test.py --opt <yml>`): loads
`experiments/<index>/model.pth`, renders every test view in `test.max_height x max_width` chunks through PAPR.evaluate on the HIP
path, reports PSNR / SSIM / LPIPS (test.py:107-110) and with `test.save_fig` (or --save) writes the reference's four PNGs per view
(test.py:120-141: predicted rgb, 16-bit depth, foreground rgb, background mask, same file names).

PSNR is the reference's formula.  SSIM is papr_amd/metrics.py: a restatement of what the reference calls
(skimage.metrics.structural_similarity, 11 x 11 uniform window), NOT pinned against scikit-image -- that package is not in this
environment.  LPIPS (AlexNet and VGG16, `lpips.LPIPS(net=..., version="0.1")` in the reference) needs pretrained weights that cannot
be fetched here: each is computed when its weight files are named (papr_amd/lpips.py: TestLPIPS) and reported as nan otherwise."""
import argparse
import os

import numpy as np
import torch

from papr_amd import get_model, load_config
from papr_amd.config import as_node, parse_overrides
from papr_amd.dataset import get_dataset
from papr_amd.lpips import TestLPIPS
from papr_amd.metrics import depth_map, ssim
from train import psnr, render_full


def main():
    ap = argparse.ArgumentParser(description="PAPR")
    ap.add_argument("--opt", type=str, default="")
    ap.add_argument("--max-views", type=int, default=-1)
    ap.add_argument("--save", action="store_true", help="write the PNGs even if test.save_fig is off")
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
    lp = {net: TestLPIPS.try_build(net, dev) for net in ("alex", "vgg")}
    for net, fn in lp.items():
        if fn is None:
            print("LPIPS (%s): weights not found, reported as nan (papr_amd/lpips.py says where they are looked for)" % net)
    save_fig = cli.save or bool(cfg["test"].get("save_fig", False))
    results = {}
    for ds in cfg["test"]["datasets"]:
        dcfg = dict(cfg["dataset"]); dcfg.update(ds)
        data = get_dataset(dcfg, ds["mode"], dev, seed=args.seed)
        n = len(data) if cli.max_views < 0 else min(len(data), cli.max_views)
        vals = {"psnr": [], "ssim": [], "lpips_alex": [], "lpips_vgg": []}
        for i in range(n):
            img, rayd, rayo, c2w = data.full_view(i)
            rgb, fg, bkg_mask, attn, sel = render_full(model, rayo, rayd, c2w, args.test.max_height, args.test.max_width, extras=True)
            m = {"psnr": psnr(rgb, img), "ssim": ssim(rgb[0].cpu().numpy(), img[0].cpu().numpy())}
            for net, fn in lp.items():
                m["lpips_" + net] = float(fn(rgb.permute(0, 3, 1, 2), img.permute(0, 3, 1, 2))) if fn is not None else float("nan")
            for k, v in m.items():
                vals[k].append(v)
            print("Test frame: %d, test_psnr: %.4f, test_ssim: %.4f, test_lpips_alex: %.4f, test_lpips_vgg: %.4f" % (i, m["psnr"], m["ssim"], m["lpips_alex"], m["lpips_vgg"]))
            if save_fig:
                from PIL import Image
                out = os.path.join(log_dir, "test", "images"); os.makedirs(out, exist_ok=True)
                tag = "codeMean%.4f-PSNR%.3f-SSIM%.4f-LPIPSA%.4f-LPIPSV%.4f.png" % (0.0, m["psnr"], m["ssim"], m["lpips_alex"], m["lpips_vgg"])
                depth = depth_map(sel, attn, rayo)                                  # (H, W) float32, scene units
                d16 = (depth / args.dataset.coord_scale * (65536 / 10)).astype(np.uint16)
                name = lambda kind: os.path.join(out, "test-%04d-%02d-%s-%s" % (i, 0, kind, tag))
                Image.fromarray((rgb[0].cpu().numpy() * 255).astype(np.uint8)).save(name("predrgb"))
                Image.fromarray(d16).save(name("depth"))
                Image.fromarray((fg[0].clamp(0, 1).cpu().numpy() * 255).astype(np.uint8)).save(name("fgrgb"))
                Image.fromarray((bkg_mask[0].cpu().numpy() * 255).astype(np.uint8)).save(name("bkgmask"))
        results[ds["name"]] = float(np.mean(vals["psnr"]))
        print("testset", ds["name"], "views", n, "avg psnr", results[ds["name"]], "avg ssim", float(np.mean(vals["ssim"])),
              "avg lpips_alex", float(np.mean(vals["lpips_alex"])), "avg lpips_vgg", float(np.mean(vals["lpips_vgg"])))
    return results


if __name__ == "__main__":
    main()
