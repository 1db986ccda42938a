"""Run-to-run determinism of one MLP through the library: python3 scripts/probes/r6_det.py <runs> <M> <n> <act> <d_in> <d_out> [env=value ...]
(the worker of tests/test_hip_chain_variants.py in fresh processes; every saved tensor compared bit for bit against the first run's)"""
import os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
runs, M, n, act, d_in, d_out = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
env = dict(os.environ, **dict(a.split("=", 1) for a in sys.argv[7:]))
ref = None
for r in range(runs):
    out = "/tmp/det_%d.pt" % r
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "chain_variants_worker.py"), out, M, n, act, d_in, d_out], env=env, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    res = torch.load(out)
    if ref is None:
        ref = res
        continue
    bad = []
    for k in sorted(res):
        for i, (a, b) in enumerate(zip(ref[k] if isinstance(ref[k], list) else [ref[k]], res[k] if isinstance(res[k], list) else [res[k]])):
            if not torch.equal(a.view(torch.int32), b.view(torch.int32)):
                bad.append("%s[%d]: %d of %d differ" % (k, i, int((a.view(torch.int32) != b.view(torch.int32)).sum()), a.numel()))
    print(" ".join(sys.argv[2:]), "run", r, "vs run 0:", "identical" if not bad else "; ".join(bad))
