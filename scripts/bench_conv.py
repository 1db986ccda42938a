#!/usr/bin/env python3
"""3x3 convolution layers of the U-Net head: papr_conv3x3_fwd against MIOpen (fp32, channels-last), forward only,
at the training patch (160 x 160) and at a full 800 x 800 image."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from papr_amd import ops
d = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for S in (160, 800):
    for name, hw, ci, co in (("inc", S, 32, 128), ("down1", S // 2, 128, 256), ("down2", S // 4, 256, 512), ("up1.conv", S // 2, 512, 256), ("up2.conv", S, 256, 128)):
        x = torch.randn(1, hw, hw, ci, device=d)
        w = torch.randn(co, ci, 3, 3, device=d) * 0.05
        b = torch.zeros(co, device=d)
        xc = x.permute(0, 3, 1, 2)                                   # NCHW view with channels-last strides
        wc = w.contiguous(memory_format=torch.channels_last)
        us_m = t(lambda: torch.relu_(torch.nn.functional.conv2d(xc, wc, b, padding=1)))
        us_h = t(lambda: ops.conv3x3_rows(x, wc, b, True))
        fl = 2.0 * hw * hw * co * 9 * ci
        print("%4d^2 %-9s %4d->%4d  MIOpen %7.1f us (%5.1f TF)   split-f16 %7.1f us (%5.1f TF)  x%.2f" % (S, name, ci, co, us_m, fl / us_m / 1e6, us_h, fl / us_h / 1e6, us_m / us_h))
print("weight gradient")
for S in (160,):
    for name, hw, ci, co in (("inc", S, 32, 128), ("down1", S // 2, 128, 256), ("down2", S // 4, 256, 512), ("up1.conv", S // 2, 512, 256), ("up2.conv", S, 256, 128)):
        x = torch.randn(1, hw, hw, ci, device=d)
        gy = torch.randn(1, hw, hw, co, device=d)
        w = (torch.randn(co, ci, 3, 3, device=d) * 0.05).contiguous(memory_format=torch.channels_last)
        us_m = t(lambda: torch.ops.aten.convolution_backward(gy.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2), w, [co], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
        us_h = t(lambda: ops.conv3x3_wgrad_rows(gy, x, True))
        fl = 2.0 * hw * hw * co * 9 * ci
        print("%4d^2 %-9s %4d->%4d  MIOpen %7.1f us (%5.1f TF)   split-f16 %7.1f us (%5.1f TF)  x%.2f" % (S, name, ci, co, us_m, fl / us_m / 1e6, us_h, fl / us_h / 1e6, us_m / us_h))
