"""PAPR_TN_TR=1 against 0 on one MLP: python3 scripts/probes/r6_tr_dbg.py <M> <n>"""
import os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M, n = sys.argv[1], sys.argv[2]
res = {}
for tr in ("0", "1"):
    out = "/tmp/trdbg_%s.pt" % tr
    env = dict(os.environ, PAPR_TN_TR=tr)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "chain_variants_worker.py"), out, M, n, "relu"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    res[tr] = torch.load(out)
for k in ("d_ws", "d_bs"):
    for i, (a, b) in enumerate(zip(res["0"][k], res["1"][k])):
        print(M, k, i, "rel max diff %.3e" % float((a - b).abs().max() / a.abs().max().clamp_min(1e-30)), "norms %.6f %.6f" % (float(a.norm()), float(b.norm())))
