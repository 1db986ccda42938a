"""Times of the U-Net layers that are not 3x3 convolutions, own kernels (unet.hip) against torch / MIOpen, at the training
shapes of the shipped configuration (160 x 160 patch): forward + backward of each layer, microseconds per call (events)."""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from papr_amd import ops

d = torch.device("cuda:0")


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def fb(f, *leaves):
    def run():
        for t in leaves:
            t.grad = None
        y = f()
        y.backward(torch.ones_like(y))
    return run


for (H, c_in, c_out) in [(40, 512, 256), (80, 256, 128)]:
    x = torch.randn(1, H, H, c_in, device=d, requires_grad=True)
    w = (torch.randn(c_in, c_out, 2, 2, device=d) * 0.05).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    b = torch.zeros(c_out, device=d, requires_grad=True)
    own = timeit(fb(lambda: ops._UpConv2x2Fn.apply(x, w, b), x, w, b))
    xt = x.detach().permute(0, 3, 1, 2).requires_grad_(True)
    ref = timeit(fb(lambda: F.conv_transpose2d(xt, w, b, stride=2), xt, w, b))
    fo = timeit(lambda: ops._UpConv2x2Fn.apply(x.detach(), w.detach(), b.detach()))
    fr = timeit(lambda: F.conv_transpose2d(xt.detach(), w.detach(), b.detach(), stride=2))
    print("upconv %dx%d %d->%d  fwd+bwd own %.1f us torch %.1f us | fwd own %.1f torch %.1f" % (H, H, c_in, c_out, own, ref, fo, fr))
for (H, Cn) in [(160, 128), (80, 256)]:
    x = torch.relu(torch.randn(1, H, H, Cn, device=d)).requires_grad_(True)
    own = timeit(fb(lambda: ops._MaxPool2Fn.apply(x), x))
    xt = x.detach().permute(0, 3, 1, 2).requires_grad_(True)
    ref = timeit(fb(lambda: F.max_pool2d(xt, 2), xt))
    print("maxpool %dx%d C %d  fwd+bwd own %.1f us torch %.1f us" % (H, H, Cn, own, ref))
x = torch.randn(1, 160, 160, 128, device=d, requires_grad=True)
w = (torch.randn(3, 128, 1, 1, device=d) * 0.05).requires_grad_(True)
b = torch.zeros(3, device=d, requires_grad=True)
own = timeit(fb(lambda: ops._Conv1x1Fn.apply(x, w, b), x, w, b))
xt = x.detach().permute(0, 3, 1, 2).requires_grad_(True)
ref = timeit(fb(lambda: F.conv2d(xt, w, b), xt, w, b))
print("conv1x1 160x160 128->3  fwd+bwd own %.1f us torch %.1f us" % (own, ref))
