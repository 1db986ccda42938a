"""Gradient error of the HIP path against the reference goldens, per tensor: max error / max |ref| and how many elements
exceed 3e-4 of the max (a few isolated outliers = activation-derivative flips at pre-activations ~ 0; a broad distribution
= arithmetic).  usage: python scripts/probes/grad_err.py [case ...]   (PAPR_GEMM_MODE=f32|h3 from the environment)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import numpy as np, torch
from conftest import case_cfg, case_rays, golden
from formula import formula_fill
from papr_amd import get_model
for tag in (sys.argv[1:] or ["chair1k", "lego1k", "variants1k", "tiny_norender"]):
    g = golden("g567_%s.npz" % tag)
    torch.manual_seed(1); np.random.seed(1)
    so = sys.stdout; sys.stdout = open(os.devnull, "w")
    m = get_model(case_cfg(tag), device="cpu"); sys.stdout = so
    formula_fill(m.state_dict())
    with torch.no_grad():
        m.points.copy_(torch.from_numpy(g["points"]))
    m = m.to("cuda")
    ro, rd, c2w = [t.cuda() for t in case_rays(tag)]
    torch.mean((m(ro, rd, c2w) - 0.5) ** 2).backward()
    named = dict(m.named_parameters())
    for key in g.files:
        if not key.startswith("grad/"):
            continue
        n = key[5:]
        if named[n].grad is None:
            continue
        ref = g[key]; got = named[n].grad.cpu().numpy()
        sc = max(np.abs(ref).max(), 1e-30)
        e = np.abs(got - ref) / sc
        print("%-14s %-8s %-55s max %.2e  >3e-4: %6d / %-8d  rms %.2e" % (tag, os.environ.get("PAPR_GEMM_MODE", "h3"), n, e.max(), int((e > 3e-4).sum()), e.size, float(np.sqrt((e ** 2).mean()))))
