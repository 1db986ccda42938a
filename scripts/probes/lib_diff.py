"""A variant build of the library against the regular one, bit for bit, on the fused-run worker of tests/test_hip_chain_variants.py:
   python scripts/probes/lib_diff.py <tag> [ENV=VALUE ...]      (scripts/probes/bin/libpapr_<tag>.so; the settings apply to the variant's runs)"""
import os, subprocess, sys, torch
tag = sys.argv[1]
extra = dict(a.split("=", 1) for a in sys.argv[2:])
def run(name, env, args):
    e = dict(os.environ); e.update(env)
    out = "/tmp/%s.pt" % name
    r = subprocess.run([sys.executable, "tests/chain_variants_worker.py", out] + args, env=e, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return torch.load(out)
bad = 0
for args in (["40000", "5", "relu"], ["45", "3", "leakyrelu"], ["20000", "4", "leakyrelu"], ["30053", "4", "relu", "141", "32"], ["30016", "4", "relu", "117", "64"]):
    a = run("a", {}, args)
    v = dict(extra); v["PAPR_HIP_LIB"] = os.path.join(os.getcwd(), "scripts/probes/bin/libpapr_%s.so" % tag)
    b = run("b", v, args)
    print("=== ", args)
    for k in sorted(a):
        va, vb = (a[k], b[k]) if isinstance(a[k], list) else ([a[k]], [b[k]])
        for i, (x, y) in enumerate(zip(va, vb)):
            if not torch.equal(x, y):
                bad += 1
                d = (x != y)
                rows = d.reshape(d.shape[0], -1).any(1).nonzero().flatten() if d.dim() > 1 else d.nonzero().flatten()
                print("  %s[%d] %s: %d of %d differ; first rows %s last %s nan %d" % (k, i, tuple(x.shape), int(d.sum()), d.numel(), rows[:6].tolist(), rows[-3:].tolist(), int(torch.isnan(y).sum())))
print("identical" if bad == 0 else "%d tensors differ" % bad)
