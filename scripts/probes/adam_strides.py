import torch
torch.manual_seed(0)
w0 = torch.randn(8, 4, 3, 3, device="cuda")
g0 = torch.randn(8, 4, 3, 3, device="cuda")
def run(fused, grad_cl):
    p = torch.nn.Parameter(w0.clone().contiguous(memory_format=torch.channels_last))
    opt = torch.optim.Adam([p], lr=1e-2, fused=fused) if fused else torch.optim.Adam([p], lr=1e-2, foreach=False)
    p.grad = g0.clone().contiguous(memory_format=torch.channels_last) if grad_cl else g0.clone()
    opt.step()
    return p.detach().clone()
ref = run(False, True)
for fused in (True,):
    for cl in (True, False):
        try:
            out = run(fused, cl)
            print("fused", fused, "grad channels_last", cl, "max diff vs reference", (out - ref).abs().max().item())
        except Exception as e:
            print("fused", fused, "grad channels_last", cl, "raised", type(e).__name__, str(e)[:100])
