"""U-Net head forward+backward time under MIOpen settings (benchmark mode, channels_last)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papr_amd.unet import SmallUNet
def run(bench, cl):
    torch.backends.cudnn.benchmark = bench
    torch.manual_seed(0)
    net = SmallUNet(32, 3).cuda()
    x = torch.randn(1, 32, 160, 160, device="cuda", requires_grad=True)
    if cl:
        net = net.to(memory_format=torch.channels_last)
    def it():
        xx = x.contiguous(memory_format=torch.channels_last) if cl else x
        y = net(xx)
        y.square().mean().backward()
    for _ in range(5): it()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): it()
    torch.cuda.synchronize()
    print("benchmark=%s channels_last=%s: %.3f ms fwd+bwd" % (bench, cl, (time.perf_counter() - t0) / 20 * 1e3))
for bench in (False, True):
    for cl in (False, True):
        run(bench, cl)
