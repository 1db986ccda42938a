"""Where do two chain_ab.py result files differ?  python scripts/probes/ab_where.py a.pt b.pt [tensor index]"""
import sys, torch
A, B = torch.load(sys.argv[1]), torch.load(sys.argv[2])
i = int(sys.argv[3]) if len(sys.argv) > 3 else 1
x, y = A["outs"][i], B["outs"][i]
bad = (x != y) | (torch.isnan(y))
rows = bad.any(1).nonzero().flatten()
print("out%d: %d bad elements in %d rows; rows mod 64 histogram:" % (i, int(bad.sum()), rows.numel()), torch.bincount(rows % 64, minlength=64).tolist())
print("rows mod 8 histogram", torch.bincount(rows % 8, minlength=8).tolist())
cols = bad.any(0).nonzero().flatten()
print("bad columns: %d; columns mod 4 histogram" % cols.numel(), torch.bincount(cols % 4, minlength=4).tolist(), "first", cols[:20].tolist())
r = int(rows[0])
c = bad[r].nonzero().flatten()
print("row %d: bad cols" % r, c[:16].tolist(), "ref", x[r, c[:6]].tolist(), "got", y[r, c[:6]].tolist())
rm = (A["rowmax"] != B["rowmax"])
print("rowmax bad per layer", rm.sum(1).tolist(), "first bad rows layer0", rm[0].nonzero().flatten()[:10].tolist(), "layer1", rm[1].nonzero().flatten()[:10].tolist())
j = int(rm[0].nonzero().flatten()[0]) if rm[0].any() else 0
print("rowmax layer0 row %d ref %g got %g ; max|out0 row| %g" % (j, A["rowmax"][0, j], B["rowmax"][0, j], A["outs"][0][j].abs().max()))
