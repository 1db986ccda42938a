python3 - <<'PY'
import os, subprocess, sys, torch
for gain in ("6", "12", "24"):
    out = "/tmp/g.pt"
    env = dict(os.environ, PAPR_GEMM_MODE="h1", PAPR_VARIANT_GAIN=gain)
    r = subprocess.run([sys.executable, "tests/chain_variants_worker.py", out, "20000", "5", "relu"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    res = torch.load(out)
    for k, v in sorted(res.items()):
        for i, t in enumerate(v if isinstance(v, list) else [v]):
            if t.dtype == torch.float32:
                print(gain, k, i, "finite" if torch.isfinite(t).all() else "NOT finite: %d nan %d inf" % (int(torch.isnan(t).sum()), int(torch.isinf(t).sum())), "max %.3g" % float(t[torch.isfinite(t)].abs().max()) if torch.isfinite(t).any() else "")
PY
