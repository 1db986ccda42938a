"""One MLP through the fused-run kernels in a fresh process (the library reads PAPR_C4_FUSED / PAPR_C4_GENERIC / PAPR_GEMM_MODE when it
loads):   python tests/chain_variants_worker.py <out.pt> <M> <n_layers> <act> [<d_in> <d_out>]

Training forward (every layer saved), data-gradient run with the saved sign words, the same without them, an inference pass;
tests/test_hip_chain_variants.py compares the files bit for bit."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(out, M, n, act, d_in=117, d_out=256):
    from papr_amd import ops
    width = 256
    gen = torch.Generator().manual_seed(M + n)
    skips = [int(v) for v in os.environ.get("PAPR_VARIANT_SKIP", "").split(",") if v]       # (the value MLP of lego.yml: skip_layers [5])
    spec = ops.MlpSpec("t", d_in, dict(n_ff_layer=n, d_ff=width, d_ff_out=d_out, norm="none", ff_act=act, ff_last_act="none", skip_layers=skips))
    d = torch.device("cuda:0")
    ws, bs = [], []
    for i in range(n):
        fi = d_in if i == 0 else width
        fo = d_out if i == n - 1 else width
        w = torch.zeros(fo, spec.layers[i]["n_in"] + (spec.ld_in if i in skips else 0))
        if i in skips:      # the skip segment [previous output | x]: columns n_in .. n_in + d_in
            w[:, spec.layers[i]["n_in"]:spec.layers[i]["n_in"] + d_in] = (torch.rand(fo, d_in, generator=gen) * 2 - 1) * (6.0 / (fi + fo)) ** 0.5
        w[:, :fi] = (torch.rand(fo, fi, generator=gen) * 2 - 1) * (6.0 / (fi + fo)) ** 0.5 * float(os.environ.get("PAPR_VARIANT_GAIN", "1"))
        ws.append(w.to(d))
        bs.append(((torch.rand(fo, generator=gen) * 2 - 1) * 0.1).to(d))
    xp = torch.zeros(M, spec.ld_in)
    xp[:, :d_in] = torch.randn(M, d_in, generator=gen)
    xd = xp.to(d)
    outs = ops.mlp_forward(spec, ws, bs, xd, M, keep=True)
    gp = torch.randn(M, spec.ld_out[-1], generator=gen)
    # gradient rows as a render produces them: magnitudes over many powers of two from row to row, and rows that are all zeros (a zero row among real
    # ones once set the slice scale of the transposing weight-gradient kernel: every real row's factor underflowed)
    gp *= torch.exp2(-torch.randint(0, 24, (M, 1), generator=gen).float())
    gp[torch.rand(M, generator=gen) < 0.05] = 0.0
    gp *= float(os.environ.get("PAPR_VARIANT_GRAD_SCALE", "1"))       # (the query MLP's gradient rows late in a run with few points: ~1e-35)
    gp = gp.to(d)
    scratch = [torch.empty((M, 256), device=d) for _ in range(2)]
    rowmax = outs.row_absmax[: n * M].clone()
    top = gp.clone()
    if os.environ.get("PAPR_VARIANT_TOP_F16") == "1":
        # the top gradient rows handed over as papr_f16_rows (what papr_attn_tail_bwd writes): what the run's own staging makes of the fp32 rows
        import f16_rows
        kind = ops.mlp_backward_takes_f16(spec, ws, bs, True)
        assert kind, "the run does not take f16 top rows"
        top = f16_rows.fill(ops.F16Rows(d, M, spec.ld_out[-1], kind == 2), gp)
    d_ws, d_bs, d_x = ops.mlp_backward(spec, ws, bs, xd, M, outs, top, scratch, True)
    d_ws2, _, d_x2 = ops.mlp_backward(spec, ws, bs, xd, M, list(outs), gp.clone(), scratch, True)      # without the saved state: fp32 masks
    inf = ops.mlp_forward(spec, ws, bs, xd, M, keep=False)[-1].clone()
    torch.cuda.synchronize()
    torch.save({"outs": [o.cpu() for o in outs], "rowmax": rowmax.cpu(), "d_ws": [t.cpu() for t in d_ws], "d_bs": [t.cpu() for t in d_bs], "d_x": d_x.cpu(),
                "d_ws2": [t.cpu() for t in d_ws2], "d_x2": d_x2.cpu(), "inf": inf.cpu()}, out)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], *[int(a) for a in sys.argv[5:7]])
