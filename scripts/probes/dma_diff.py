import os, subprocess, sys, torch
ROOT = os.getcwd()
def run(name, env, args):
    e = dict(os.environ); e.update(env)
    out = "/tmp/%s.pt" % name
    r = subprocess.run([sys.executable, "tests/chain_variants_worker.py", out] + args, env=e, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return torch.load(out)
for args in (["30053", "4", "relu", "141", "32"], ["30016", "4", "relu", "141", "32"], ["30053", "4", "relu", "141", "64"], ["30053", "4", "relu", "117", "256"]):
    a = run("a", {}, args); b = run("b", {"PAPR_C4_DMA": "1"}, args)
    print("=== ", args)
    for k in sorted(a):
        va, vb = (a[k], b[k]) if isinstance(a[k], list) else ([a[k]], [b[k]])
        for i, (x, y) in enumerate(zip(va, vb)):
            if not torch.equal(x, y):
                d = (x != y)
                rows = d.reshape(d.shape[0], -1).any(1).nonzero().flatten() if d.dim() > 1 else d.nonzero().flatten()
                print("  %s[%d] %s: %d of %d differ; first rows %s last %s nan %d" % (k, i, tuple(x.shape), int(d.sum()), d.numel(), rows[:6].tolist(), rows[-3:].tolist(), int(torch.isnan(y).sum())))
