"""chain4.hip against chain3.hip key by key (tests/chain_variants_worker.py outputs): which tensors differ and how."""
import os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M, n, act = (sys.argv[1:4] + ["40000", "5", "relu"][len(sys.argv) - 1:])[:3]
res = {}
for v in ("3", "4"):
    out = "/tmp/c4diff_%s.pt" % v
    e = dict(os.environ, PAPR_CHAIN=v)
    if v == "3": e.pop("PAPR_C4_FUSED", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "chain_variants_worker.py"), out, M, n, act], env=e, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    res[v] = torch.load(out)
def flat(d):
    for k, v in sorted(d.items()):
        if isinstance(v, list):
            for i, t in enumerate(v): yield "%s[%d]" % (k, i), t
        else: yield k, v
for (k, a), (_, b) in zip(flat(res["3"]), flat(res["4"])):
    nd = int((a != b).sum())
    extra = ""
    if nd:
        d = (a - b).abs()
        idx = (a != b).nonzero()
        extra = " max|diff| %g  max|ref| %g  first %s  rows differing %d" % (float(d.max()), float(a.abs().max()), idx[0].tolist(), len(set(idx[:, 0].tolist())))
    print("%-10s %-18s differ %d of %d%s" % (k, tuple(a.shape), nd, a.numel(), extra))
import collections
a, b = res["3"]["rowmax"], res["4"]["rowmax"]
Mi = int(M)
idx = (a != b).nonzero()[:, 0]
print("rowmax: differing entries per layer", collections.Counter((idx // Mi).tolist()))
r = idx % Mi
print("row %% 128 histogram (bucket of 16):", sorted(collections.Counter(((r % 128) // 16).tolist()).items()))
print("tile-pair index histogram (first 20):", sorted(collections.Counter((r // 128).tolist()).items())[:20])
for j in idx[:8].tolist():
    print("  entry", j, "layer", j // Mi, "row", j % Mi, "chain3", float(a[j]), "chain4", float(b[j]))
o = res["3"]["outs"]
for j in idx[:4].tolist():
    l, row = j // Mi, j % Mi
    if l >= 1:
        print("  true max of outs[%d][%d] = %g" % (l - 1, row, float(o[l - 1][row].abs().max())), "per 32-column block:", [float(o[l - 1][row, 32 * w:32 * w + 32].abs().max()) for w in range(8)])
