"""A/B of the two fused-run kernels (PAPR_CHAIN=1: chain.hip, 2: chain2.hip) on one MLP: every saved activation, the
weight / bias / input gradients.   usage: python scripts/probes/chain_ab.py run <out.pt> | cmp <a.pt> <b.pt>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

def run(out):
    from papr_amd import ops
    M, d_in, width, d_out, n, act = 1000, 117, 256, 256, 5, "relu"
    gen = torch.Generator().manual_seed(M)
    spec = ops.MlpSpec("t", d_in, dict(n_ff_layer=n, d_ff=width, d_ff_out=d_out, norm="none", ff_act=act, ff_last_act="none", skip_layers=[]))
    d = torch.device("cuda:0")
    ws, bs = [], []
    for i in range(n):
        fi = d_in if i == 0 else width
        fo = d_out if i == n - 1 else width
        w = torch.zeros(fo, spec.layers[i]["n_in"]); w[:, :fi] = (torch.rand(fo, fi, generator=gen) * 2 - 1) * (6.0 / (fi + fo)) ** 0.5
        ws.append(w.to(d)); bs.append(((torch.rand(fo, generator=gen) * 2 - 1) * 0.1).to(d))
    xp = torch.zeros(M, spec.ld_in); xp[:, :d_in] = torch.randn(M, d_in, generator=gen)
    xd = xp.to(d)
    outs = ops.mlp_forward(spec, ws, bs, xd, M, keep=True)
    gp = torch.randn(M, spec.ld_out[-1], generator=gen).to(d)
    scratch = [torch.empty((M, 256), device=d) for _ in range(2)]
    saved = outs.row_absmax.clone()
    d_ws, d_bs, d_x = ops.mlp_backward(spec, ws, bs, xd, M, outs, gp.clone(), scratch, True)
    d_ws2, d_bs2, d_x2 = ops.mlp_backward(spec, ws, bs, xd, M, list(outs), gp.clone(), scratch, True)      # without the saved state: fp32 masks
    torch.cuda.synchronize()
    from papr_amd import hip
    import ctypes
    if hasattr(hip.lib(), "papr_chain2_dbg_read"):
        buf = (ctypes.c_uint * 16)()
        hip.lib().papr_chain2_dbg_read(buf)
        print("debug counters (rows 0-3 changed on re-read, scales changed, batches):", list(buf))
    torch.save({"d_ws2": [t.cpu() for t in d_ws2], "d_x2": d_x2.cpu(), "outs": [o.cpu() for o in outs], "rowmax": saved[: n * M].cpu().view(n, M), "d_ws": [t.cpu() for t in d_ws], "d_bs": [t.cpu() for t in d_bs], "d_x": d_x.cpu()}, out)

def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    x, y = A["d_x"], B["d_x"]
    bad = ((x - y).abs().max(1).values > 1e-6 * x.abs().max()).nonzero().flatten().tolist()
    for r in bad[:12]:
        nz = x[r].abs() > 1e-3
        print("d_x row %d (row %d of its tile): ratio b/a of its columns: min %.4f max %.4f" % (r, r % 64, (y[r][nz] / x[r][nz]).min(), (y[r][nz] / x[r][nz]).max()))
    def rep(name, x, y):
        e = (x - y).abs()
        print("%-12s max|a| %.3e  max err %.3e  mismatched %d / %d  first bad row %s" % (name, x.abs().max(), e.max(), int((e > 1e-6 * x.abs().max()).sum()), e.numel(),
              (e.reshape(e.shape[0], -1).max(1).values > 1e-6 * x.abs().max()).nonzero()[:5].flatten().tolist()))
    for i, (x, y) in enumerate(zip(A["outs"], B["outs"])): rep("out%d" % i, x, y)
    rep("rowmax", A["rowmax"], B["rowmax"])
    for i, (x, y) in enumerate(zip(A["d_ws"], B["d_ws"])): rep("dW%d" % i, x, y)
    for i, (x, y) in enumerate(zip(A["d_bs"], B["d_bs"])): rep("db%d" % i, x[:, None], y[:, None])
    rep("d_x", A["d_x"], B["d_x"])
    for i, (x, y) in enumerate(zip(A["d_ws2"], B["d_ws2"])): rep("dW%d nosave" % i, x, y)
    rep("d_x nosave", A["d_x2"], B["d_x2"])

if __name__ == "__main__":
    run(sys.argv[2]) if sys.argv[1] == "run" else cmp(sys.argv[2], sys.argv[3])
