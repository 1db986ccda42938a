"""The reduced-precision mode of the fused MLP runs (PAPR_GEMM_MODE=h1: one f16 product per fp32 product, chain4.hip) in a
fresh process, because the library reads the mode when it loads:   python tests/h1_worker.py <case tag> <out.json>

Renders the golden case and takes its gradients exactly like tests/test_hip_model.py does in the parity mode, and writes the
errors against the reference golden (tests/golden/g567_<tag>.npz) for tests/test_hip_h1.py to judge with the mode's tolerance.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(tag, out):
    assert os.environ.get("PAPR_GEMM_MODE") == "h1" or os.environ.get("PAPR_WORKER_ANY_MODE") == "1"      # (scripts/probes/r6_det_model.sh runs it in the default mode)
    from conftest import case_cfg, case_rays, golden
    from formula import formula_fill
    from papr_amd import get_model
    g = golden("g567_%s.npz" % tag)
    torch.manual_seed(1)
    np.random.seed(1)
    m = get_model(case_cfg(tag), device="cpu")
    formula_fill(m.state_dict())
    with torch.no_grad():
        m.points.copy_(torch.from_numpy(g["points"]))
    m = m.to("cuda")
    ro, rd, c2w = [t.to("cuda") for t in case_rays(tag)]
    with torch.no_grad():
        fused, attn = m.evaluate(ro, rd, c2w)
    m.clear_grad()
    rgb = m(ro, rd, c2w)
    loss = torch.mean((rgb - 0.5) ** 2)
    loss.backward()
    res = {"rgb": float(np.abs(rgb.detach().cpu().numpy() - g["rgb"]).max()),
           "fused": float(np.abs(fused.squeeze(-2).cpu().numpy() - g["fused"]).max()), "fused_scale": float(np.abs(g["fused"]).max()),
           "loss": float(loss), "loss_ref": float(g["loss"]), "grads": {}}
    named = dict(m.named_parameters())
    for key in g.files:
        if key.startswith("grad/") and named[key[5:]].grad is not None:
            ref = g[key].astype(np.float64)
            got = named[key[5:]].grad.cpu().numpy().astype(np.float64)
            scale = np.abs(ref).max()
            if scale > 0:
                res["grads"][key[5:]] = {"rms_rel": float(np.sqrt(np.mean((got - ref) ** 2)) / scale), "max_rel": float(np.abs(got - ref).max() / scale),
                                         "finite": bool(np.isfinite(got).all())}
    # (a fingerprint of the weight gradients, for the test that the f16-row path and the fp32-row path are two different computations)
    res["digest"] = float(sum(np.abs(named[k[5:]].grad.cpu().numpy().astype(np.float64)).sum() for k in g.files
                              if k.startswith("grad/") and named[k[5:]].grad is not None and "attn" in k))
    from papr_amd import ops
    res["tail_f16_rows"] = ops.TAIL_F16_ROWS_USED      # how many gradient-row sets left papr_attn_tail_bwd as papr_f16_rows
    json.dump(res, open(out, "w"))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
