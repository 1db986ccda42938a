"""Small U-Net render head (32 -> 128 -> 256 -> 512 -> 256 -> 128 -> 3).

Counterpart of the reference's SmallUNet (models/unet.py:182-258) in its only shipped
configuration (single conv per stage, transposed-conv upsampling, no normalisation, no affine
modulation); it keeps the reference's module attribute names so that state-dict keys are
interchangeable.  On the device every layer runs on this library's kernels over NHWC maps: the 3x3
convolutions on conv.hip, max-pooling, the transposed convolutions and the 1x1 output layer on
unet.hip.  Where the map's height and width are multiples of 4 (training patches, whole 800 x 800 views) the
whole network is ONE library call each way (papr_small_unet_fwd / _bwd, csrc/small_unet.hip: one launch splits every 3x3 weight, tensor maxima
come from the producing kernels, skip concatenations are written in place, ReLU masks and skip-gradient sums ride in the seam kernels;
PAPR_UNET_NET=0: layer by layer, which other sizes take anyway).  With `use_amp: true` as well: the reference wraps this module in fp16
autocast there (models/unet.py:212); here the same kernels run with fp32 maps and fp32 accumulation -- on the whole-network path with ONE f16
product per fp32 product (the reference's operand precision under autocast; PAPR_UNET_AMP_ONE=0: three, as without use_amp), layer by layer
with three.

No fallback on the device (round 6): a CUDA tensor reaches torch's own layers (nn.Conv2d, MaxPool2d, ConvTranspose2d, autocast) ONLY behind the one debug
switch PAPR_DEBUG_TORCH_HEAD (papr_amd/debug.py: conv | rest | wgrad | autocast | all, announced on stderr when it engages); a shape outside the own
kernels' constraints raises NotImplementedError by name.  On the CPU the modules are plain torch: the float64 reference of the GPU tests, not a product path.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .debug import own_or_raise, torch_head

_WHOLE_NET = os.environ.get("PAPR_UNET_NET", "1") == "1"       # (0: layer by layer -- one autograd function per layer, torch glue between them; A/B)
_AMP_ONE = os.environ.get("PAPR_UNET_AMP_ONE", "1") == "1"     # under use_amp, whole-network path: one f16 product per fp32 product (0: three, as without use_amp; A/B)


class ConvStage(nn.Module):
    """Conv2d(3x3, padding 1) + ReLU (the reference's DoubleConv with single=True, models/unet.py:16-33).  On the device it runs on the
    split-f16 implicit-GEMM kernels over the NHWC map (papr_conv3x3_fwd: forward and data-gradient; papr_conv3x3_wgrad: weight gradient);
    channel counts they have no form for raise.  A CPU tensor takes torch's convolution (the tests' float64 reference)."""

    def __init__(self, c_in, c_out):
        super().__init__()
        self.double_conv = nn.Sequential(nn.Conv2d(c_in, c_out, kernel_size=3, padding=1), nn.ReLU(inplace=True))

    def forward(self, x):
        conv = self.double_conv[0]
        if _own(x, conv.in_channels % 32 == 0 and conv.out_channels % 32 == 0, "conv", "Conv2d(%d, %d, 3x3)" % (conv.in_channels, conv.out_channels),
                "input and output channels multiples of 32"):
            from .ops import _Conv3x3Fn
            rows = x.permute(0, 2, 3, 1).contiguous()             # (no copy when the map is channels-last already)
            return _Conv3x3Fn.apply(rows, conv.weight, conv.bias, True).permute(0, 3, 1, 2)
        return self.double_conv(x)


def _plain_f32(x):
    return x.dtype == torch.float32 and not torch.is_autocast_enabled()


def _own(x, ok, part, layer, constraint):
    """Which way a layer goes.  True: this library's kernel.  False: torch's layer -- for a CPU tensor (the tests' reference, not a product path) and, on the
    device, ONLY behind PAPR_DEBUG_TORCH_HEAD (papr_amd/debug.py; `autocast` hands every layer over, the reference's arrangement).  A device tensor
    outside the own kernel's constraints (`ok` false, a dtype other than fp32, a caller's autocast region) raises NotImplementedError by name."""
    if not x.is_cuda:
        return False
    if torch.is_autocast_enabled() and torch_head("autocast"):
        return False
    return own_or_raise(_plain_f32(x) and ok, part, "%s on a %s map%s" % (layer, str(x.dtype).replace("torch.", ""), " under autocast" if torch.is_autocast_enabled() else ""),
                        "fp32 maps outside autocast, " + constraint)


def _own_rest(x, ok, layer, constraint):
    return _own(x, ok, "rest", layer, constraint)


class DownStage(nn.Module):
    """MaxPool2d(2) + ConvStage (models/unet.py:36-49); the pooling on papr_maxpool2_fwd / _bwd."""

    def __init__(self, c_in, c_out):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), ConvStage(c_in, c_out))

    def forward(self, x):
        if _own_rest(x, x.shape[1] % 4 == 0 and x.shape[2] >= 2 and x.shape[3] >= 2, "MaxPool2d(2) over %s" % (tuple(x.shape),),
                     "channels a multiple of 4 and a map of at least 2 x 2"):
            from .ops import _MaxPool2Fn
            pooled = _MaxPool2Fn.apply(x.permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)
            return self.maxpool_conv[1](pooled)
        return self.maxpool_conv(x)


class UpStage(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.up = nn.ConvTranspose2d(c_in, c_in // 2, kernel_size=2, stride=2)
        self.conv = ConvStage(c_in, c_out)

    def forward(self, low, skip):
        if _own_rest(low, self.up.in_channels % 64 == 0 and self.up.out_channels % 64 == 0,
                     "ConvTranspose2d(%d, %d, 2x2)" % (self.up.in_channels, self.up.out_channels), "input and output channels multiples of 64"):
            from .ops import _UpConv2x2Fn                      # (papr_upconv2x2_*: forward and both gradients one launch each)
            low = _UpConv2x2Fn.apply(low.permute(0, 2, 3, 1).contiguous(), self.up.weight, self.up.bias).permute(0, 3, 1, 2)
        else:
            low = self.up(low)
        dy, dx = skip.shape[2] - low.shape[2], skip.shape[3] - low.shape[3]
        if dx or dy:                                   # (F.pad with nothing to pad still copies the map, forward and backward)
            low = F.pad(low, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
        return self.conv(torch.cat([skip, low], dim=1))


class Head(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.conv = nn.Conv2d(c_in, c_out, kernel_size=1)

    def forward(self, x):
        c_in, c_out = self.conv.in_channels, self.conv.out_channels
        if _own_rest(x, c_out <= 4 and c_in in (32, 64, 128, 256), "Conv2d(%d, %d, 1x1)" % (c_in, c_out), "at most 4 classes and 32 / 64 / 128 / 256 input channels"):
            from .ops import _Conv1x1Fn
            return _Conv1x1Fn.apply(x.permute(0, 2, 3, 1).contiguous(), self.conv.weight, self.conv.bias).permute(0, 3, 1, 2)
        return self.conv(x)


class SmallUNet(nn.Module):
    def __init__(self, n_channels, n_classes, use_amp=False, amp_dtype=torch.float16, last_act="none"):
        super().__init__()
        self.use_amp, self.amp_dtype = use_amp, amp_dtype
        self.inc = ConvStage(n_channels, 128)
        self.down1 = DownStage(128, 256)
        self.down2 = DownStage(256, 512)
        self.up1 = UpStage(512, 256)
        self.up2 = UpStage(256, 128)
        self.outc = Head(128, n_classes)
        # `small_unet.last_act` (reference models/unet.py:205,253: activation_func(last_act) on the logits, default arguments): the same table as the
        # model's output activation (papr_amd/activations.py, pinned by G16), an elementwise function of the head's (N, classes, H, W) result --
        # the state-dict keys are the reference's (`last_act.a` for the names with a shape constant).  (PReLU's 128 channels cannot meet 3 classes in the
        # reference either.)
        from .activations import output_activation
        self.last_act = output_activation(last_act)
        # The feature map arrives as (N, H, W, C) rows; channels-last weights are what the own kernels read in place (the state dict is
        # unchanged: same keys, shapes and values, only the strides differ).
        self.to(memory_format=torch.channels_last)

    def forward(self, x, gamma=None, beta=None):
        # the whole network as one library call each way (papr_small_unet_fwd / _bwd) where its shapes allow: two exact poolings (H, W multiples
        # of 4), input channels a multiple of 32, biases present; otherwise layer by layer below
        on_torch = x.is_cuda and (torch_head("conv") or torch_head("rest") or (self.use_amp and torch_head("autocast")))       # (debug switch: layer by layer below)
        if (_WHOLE_NET and x.is_cuda and not on_torch and _plain_f32(x) and x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0 and x.shape[2] >= 4
                and x.shape[3] >= 4 and x.shape[1] % 32 == 0 and self.outc.conv.out_channels <= 4 and x.shape[0] * x.shape[2] * x.shape[3] * 2048 < 2 ** 31):
            from .ops import small_unet_rows
            # use_amp: the reference autocasts this module to fp16 (models/unet.py:212: f16 operands, f16 maps); here f16 OPERANDS (one product per
            # fp32 product, fp32 accumulation) and fp32 maps -- pinned to the reference's own AMP output by G17 (tests/test_hip_amp_golden.py)
            return self.last_act(small_unet_rows(x.permute(0, 2, 3, 1).contiguous(), self, one_product=self.use_amp and _AMP_ONE).permute(0, 3, 1, 2))
        # layer by layer: other map sizes on the same kernels (each layer raises by name where its own kernel has no form); a CPU tensor: plain torch
        with torch.autocast(device_type="cuda", dtype=self.amp_dtype, enabled=self.use_amp and x.is_cuda and torch_head("autocast")):
            x1 = self.inc(x)
            x2 = self.down1(x1)
            x3 = self.down2(x2)
            y = self.up2(self.up1(x3, x2), x1)
            return self.last_act(self.outc(y))


class MLPGenerator(nn.Module):
    """Per-pixel MLP render head on the HIP MLP kernels -- counterpart of the reference's MLPGenerator
    (models/renderer.py:6-17: an MLP, models/mlp.py:12-59, over the channels of every pixel).  Same parameter names
    (`mlp.model.{2i+1}.weight/bias`), same options as far as the kernels' epilogues go: relu / leakyrelu / none,
    skip layers, bias; weight-norm, half / residual layers and the exotic activations raise."""

    def __init__(self, inp_dim, num_layers, num_channels, out_dim, act_type="leakyrelu", last_act_type="none", use_wn=False,
                 skip_layers=(), bias=True, half_layers=(), residual_layers=(), residual_dims=()):
        super().__init__()
        from .model import _MlpParams
        from .ops import MlpSpec
        if use_wn or not bias or half_layers or residual_layers or residual_dims:
            raise NotImplementedError("papr_amd: MLPGenerator supports use_wn=false, bias=true, no half / residual layers")
        ecfg = dict(n_ff_layer=num_layers, d_ff=num_channels, d_ff_out=out_dim, norm="none", ff_act=act_type,
                    ff_last_act=last_act_type, skip_layers=list(skip_layers))
        self.spec = MlpSpec("generator", inp_dim, ecfg)     # raises for activations without a HIP epilogue
        self.mlp = _MlpParams(inp_dim, ecfg)
        self.out_dim = out_dim

    def forward(self, x, residuals=(), gamma=None, beta=None):      # (N, C, H, W) -> (N, out, H, W)
        from .ops import mlp_rows
        N, C, H, W = x.shape
        lin = self.mlp.linears()
        rows = x.permute(0, 2, 3, 1).reshape(-1, C)
        y = mlp_rows(self.spec, rows, [m.weight for m in lin], [m.bias for m in lin])
        return y.reshape(N, H, W, self.out_dim).permute(0, 3, 1, 2)


def get_generator(gcfg, in_c, out_c, use_amp=False, amp_dtype=torch.float16):
    """Counterpart of models/renderer.py:21-34."""
    if gcfg["type"] == "small-unet":
        o = gcfg["small_unet"]
        if o["bilinear"] or not o["single"] or o["norm"] != "none" or o["affine_layer"] >= 0:
            raise NotImplementedError("papr_amd: small-unet with bilinear / double-conv / norm / affine_layer is not built (the shipped variant: transposed-conv, "
                                      "single, no norm, no affine; any last_act)")
        return SmallUNet(in_c, out_c, use_amp=use_amp, amp_dtype=amp_dtype, last_act=o.get("last_act", "none"))
    if gcfg["type"] == "mlp":
        if "mlp" not in gcfg:
            raise KeyError("models.renderer.generator.mlp: option block missing (the reference reads num_layers, num_channels, "
                           "act_type, ... from it, models/renderer.py:26-31; its default.yml ships none)")
        o = gcfg["mlp"]
        if float(o.get("act_a", 1.0)) != 1.0 or float(o.get("act_b", 1.0)) != 1.0 or o.get("act_trainable", False):
            raise NotImplementedError("papr_amd: MLPGenerator activations take no parameters (act_a / act_b / act_trainable)")
        return MLPGenerator(in_c, o["num_layers"], o["num_channels"], out_c, act_type=o["act_type"], last_act_type=o["last_act_type"],
                            use_wn=o.get("use_wn", False), skip_layers=o.get("skip_layers", []) or [], bias=o.get("bias", True),
                            half_layers=o.get("half_layers", []) or [], residual_layers=o.get("residual_layers", []) or [],
                            residual_dims=o.get("residual_dims", []) or [])
    raise NotImplementedError("generator type [%s] is not supported by papr_amd" % gcfg["type"])
