"""Small U-Net render head (32 -> 128 -> 256 -> 512 -> 256 -> 128 -> 3) on torch / MIOpen.

Counterpart of the reference's SmallUNet (models/unet.py:182-258) in its only shipped
configuration (single conv per stage, transposed-conv upsampling, no normalisation, no affine
modulation).  It is convolutional over the whole patch rather than per ray, so it is not part of
the hand-written per-ray kernels; it keeps the reference's module
attribute names so that state-dict keys are interchangeable.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ConvStage(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.double_conv = nn.Sequential(nn.Conv2d(c_in, c_out, kernel_size=3, padding=1), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.double_conv(x)


class DownStage(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), ConvStage(c_in, c_out))

    def forward(self, x):
        return self.maxpool_conv(x)


class UpStage(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.up = nn.ConvTranspose2d(c_in, c_in // 2, kernel_size=2, stride=2)
        self.conv = ConvStage(c_in, c_out)

    def forward(self, low, skip):
        low = self.up(low)
        dy, dx = skip.shape[2] - low.shape[2], skip.shape[3] - low.shape[3]
        low = F.pad(low, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return self.conv(torch.cat([skip, low], dim=1))


class Head(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.conv = nn.Conv2d(c_in, c_out, kernel_size=1)

    def forward(self, x):
        return self.conv(x)


class SmallUNet(nn.Module):
    def __init__(self, n_channels, n_classes, use_amp=False, amp_dtype=torch.float16):
        super().__init__()
        self.use_amp, self.amp_dtype = use_amp, amp_dtype
        self.inc = ConvStage(n_channels, 128)
        self.down1 = DownStage(128, 256)
        self.down2 = DownStage(256, 512)
        self.up1 = UpStage(512, 256)
        self.up2 = UpStage(256, 128)
        self.outc = Head(128, n_classes)

    def forward(self, x, gamma=None, beta=None):
        with torch.autocast(device_type="cuda", dtype=self.amp_dtype, enabled=self.use_amp and x.is_cuda):
            x1 = self.inc(x)
            x2 = self.down1(x1)
            x3 = self.down2(x2)
            y = self.up2(self.up1(x3, x2), x1)
            return self.outc(y)


def get_generator(gcfg, in_c, out_c, use_amp=False, amp_dtype=torch.float16):
    """Counterpart of models/renderer.py:21-34."""
    if gcfg["type"] == "small-unet":
        o = gcfg["small_unet"]
        if o["bilinear"] or not o["single"] or o["norm"] != "none" or o["affine_layer"] >= 0 or o["last_act"] != "none":
            raise NotImplementedError("papr_amd: only the shipped small-unet variant (transposed-conv, single, no norm/affine) is built")
        return SmallUNet(in_c, out_c, use_amp=use_amp, amp_dtype=amp_dtype)
    raise NotImplementedError("generator type [%s] is not supported by papr_amd" % gcfg["type"])
