"""Host-side orchestration of the HIP render path (kNN -> features -> embedding MLPs -> attention tail).

`RenderPath` owns the static geometry of the path (widths, paddings, layer tables) derived from the
config; `render_rays` is the differentiable entry point used by `papr_amd.model.PAPR`.  All device
work goes through libpapr_hip.so (papr_amd/hip.py); torch is used for memory, streams and for the
tiny weight preparation (LayerNorm-affine folding, zero padding, transposes), which keeps those
pieces inside ordinary autograd.

Reference call sites this file stands in for: PAPR._get_points / _get_kqv / evaluate / forward
(models/model.py:312-333, 396-437, 462-545) and ProximityAttention.forward (models/attn.py:241-252).
"""
import ctypes as C

import os
import torch
import torch.nn.functional as F

from . import hip


def mlp_mode(one_product, fp32_rows_inside=False):
    """The `mode` argument of papr_mlp_fwd / papr_mlp_bwd (include/papr_hip.h: PAPR_MLP_*) for one call.  The library itself reads no
    environment: PAPR_GEMM_MODE (h3 default | h3_f16rows | h1 | layers | dgrad | fwd | f32: A/B and benchmark switch) is THIS module's, and a
    `use_amp: true` model asks for the one-product arithmetic call by call (on top of the default mode only, so that the A/B modes
    stay what they say)."""
    name = os.environ.get("PAPR_GEMM_MODE", "h3")
    if name == "h3" and one_product:
        name = "h1"
    # round 6: the parity mode's fused runs keep the rows their weight gradients read as f16 rows (PAPR_MLP_H3_F16ROWS: forward and data-gradient
    # bit for bit those of PAPR_MLP_H3, every golden bar unchanged, 10.4 -> 8.9 ms per step).  PAPR_H3_ROWS=f32: fp32 rows (A/B)
    if name == "h3" and not fp32_rows_inside and os.environ.get("PAPR_H3_ROWS", "f16") != "f32":
        name = "h3_f16rows"
    if name == "h1" and (os.environ.get("PAPR_H1_ROWS", "") == "f32" or fp32_rows_inside):      # (A/B: fp32 rows between a run and its weight gradients;
        name = "h1_f32rows"                                                                       # fp32_rows_inside: a run whose INNER rows the host reads)
    if name == "h3_f16rows" and fp32_rows_inside:       # (named explicitly: a run whose inner rows the host reads keeps fp32 rows all the same)
        name = "h3"
    return hip.MLP_MODES[name]


def _pad4(n):
    return (n + 3) // 4 * 4


def _pad32(n):
    return (n + 31) // 32 * 32


class MlpSpec:
    """Static description of one embedding MLP (reference models/mlp.py:12-45 as configured by
    models/attn.py:152-163)."""

    one_product = False        # True: this MLP's library calls run in the one-product (h1) arithmetic (mode argument PAPR_MLP_H1)

    def __init__(self, name, d_in, ecfg):
        self.name = name
        self.d_in = d_in
        self.ld_in = _pad4(d_in)
        self.n_layer = int(ecfg["n_ff_layer"])
        self.width = int(ecfg["d_ff"])
        self.d_out = int(ecfg["d_ff_out"])
        self.norm = ecfg["norm"]
        self.skip_layers = list(ecfg.get("skip_layers", []) or [])
        # half_layers (models/mlp.py:27-30): layer i in the list takes half the width as input -- layer i - 1 produces only that much
        self.half_layers = [int(i) for i in (ecfg.get("half_layers", []) or [])]
        if ecfg.get("residual_layers"):
            raise NotImplementedError("papr_amd: %s.residual_layers is not supported by the HIP path" % name)
        # (use_wn: the weights reach the kernels as g v / |v|, formed in torch ops inside autograd -- papr_amd/model.py: effective_weight)
        if ecfg.get("residual_ff", False) and d_in == self.d_out:
            raise NotImplementedError("papr_amd: %s.residual_ff is not supported by the HIP path" % name)
        if float(ecfg.get("dropout_ff", 0.0)) != 0.0:
            raise NotImplementedError("papr_amd: %s.dropout_ff > 0 is not supported by the HIP path" % name)
        if self.norm not in ("layernorm", "none"):
            raise ValueError("Invalid attention norm type")
        for a in (ecfg["ff_act"], ecfg["ff_last_act"]):
            if a.lower() not in hip.ACT:
                raise NotImplementedError("papr_amd: activation '%s' (%s) has no HIP epilogue (relu/leakyrelu/none)" % (a, name))
        self.act = hip.ACT[ecfg["ff_act"].lower()]
        self.last_act = hip.ACT[ecfg["ff_last_act"].lower()]
        # (<= 256 and multiples of 32: the fused runs; wider or odd layers run layer by layer on the split-f16 / fp32 GEMMs, their weight gradients
        # in 256 x 256 blocks -- 1024: the row kernels' and the LayerNorm fold's limit)
        if self.width % 4 or self.width > 1024 or self.d_out > 1024:
            raise NotImplementedError("papr_amd: %s widths must be multiples of 4 and <= 1024" % name)
        # per-layer geometry
        self.layers = []
        for i in range(self.n_layer):
            raw_in = self.d_in if i == 0 else self.width
            n_out = self.d_out if i == self.n_layer - 1 else self.width
            if i + 1 in self.half_layers:
                n_out //= 2
            if i in self.half_layers:
                raw_in //= 2
            if (i in self.half_layers and (i == 0 or raw_in % 32)) or (i + 1 in self.half_layers and n_out % 32):
                raise NotImplementedError("papr_amd: %s.half_layers needs halved widths that are multiples of 32 (layer %d: %d -> %d)" % (name, i, raw_in, n_out))
            n_in = self.ld_in if i == 0 else raw_in
            skip = i in self.skip_layers
            self.layers.append(dict(n_in=n_in, n_out=n_out, n_out_pad=_pad32(n_out) if i == self.n_layer - 1 else n_out,
                                    raw_in=raw_in, skip=skip,
                                    act=self.last_act if i == self.n_layer - 1 else self.act))
        self.ld_out = [l["n_out_pad"] for l in self.layers]


_zero_cache = {}


def _zeros(dev, shape):
    """A constant block of fp32 zeros on `dev` (read-only operands such as padding rows and absent biases: made once, not filled every step)."""
    key = (dev.type, dev.index, tuple(shape))
    z = _zero_cache.get(key)
    if z is None:
        z = _zero_cache[key] = torch.zeros(shape, device=dev, dtype=torch.float32)
    return z


def _const(dev, n, value):
    key = (dev.type, dev.index, int(n), float(value))
    z = _zero_cache.get(key)
    if z is None:
        z = _zero_cache[key] = torch.full((n,), value, device=dev, dtype=torch.float32)
    return z


def conv3x3_rows(x, weight, bias, relu, transposed=False, want_max=False):
    """x (B, H, W, C_in) fp32 contiguous, weight = the reference's (C_out, C_in, 3, 3) Conv2d parameter in any memory format
    -> (B, H, W, C_out): 3x3 convolution, stride 1, zero padding 1, on the split-f16 implicit-GEMM kernel.  transposed:
    the layer's data-gradient (x = d_out: channels and taps of the weight exchanged / mirrored inside the kernel's weight
    split).  No autograd here (see _Conv3x3Fn)."""
    if not x.is_cuda:
        raise RuntimeError("papr_amd: the convolution kernel runs only on a ROCm device (HIP); there is no CPU fallback")
    B, H, W, c_in = x.shape
    sn, sc, sky, skx = weight.stride()
    c_out = weight.shape[0]
    if transposed:
        c_out, sn, sc = weight.shape[1], sc, sn
    assert weight.shape[1 if not transposed else 0] == c_in and x.is_contiguous()
    lib = hip.lib()
    out = torch.empty((B, H, W, c_out), device=x.device, dtype=torch.float32)
    size = lib.papr_conv3x3_workspace_bytes(B, H, W, c_in, c_out)
    key = (x.device.type, x.device.index, "conv")
    ws, calls = _ws_cache.get(key, (None, 0))
    if ws is None or ws.numel() * 4 < size:
        ws, calls = torch.empty((size + 3) // 4, device=x.device, dtype=torch.float32), 0
        ws[:64].zero_()                                      # the 64 maximum slots (include/papr_hip.h)
    _ws_cache[key] = (ws, calls + 1)
    hip.check(lib.papr_conv3x3_fwd(hip.ptr(x), B, H, W, c_in, C.c_void_p(weight.data_ptr()), sn, sc, sky, skx, 1 if transposed else 0,
                                   hip.ptr(bias), c_out, 1 if relu else 0, hip.ptr(out), hip.ptr(ws), calls % 64, hip.stream_ptr()), "papr_conv3x3_fwd")
    if want_max:                                             # where this call left max |x| (valid for the next 31 calls on this workspace)
        return out, (ws, calls)
    return out


def _conv_max_ptr(token, dev):
    """Device address of the maximum a conv3x3_rows call left, or None once its slot has been given out again."""
    ws, calls = token
    cur_ws, cur_calls = _ws_cache.get((dev.type, dev.index, "conv"), (None, 0))
    if cur_ws is not ws or cur_calls - calls >= 32:
        return None
    return C.c_void_p(ws.data_ptr() + 4 * (calls % 64))


def conv3x3_wgrad_rows(d_y, x, want_bias=True, d_y_max=None, x_max=None):
    """d_y (B, H, W, C_out), x (B, H, W, C_in) fp32 contiguous -> (weight gradient (C_out, C_in, 3, 3) in channels-last memory
    format, bias gradient (C_out,) or None)  (papr_conv3x3_wgrad: pixels reduced chunk by chunk, chunks added in a fixed order)."""
    B, H, W, c_out = d_y.shape
    c_in = x.shape[3]
    lib = hip.lib()
    size = lib.papr_conv3x3_wgrad_workspace_bytes(B, H, W, c_in, c_out)
    key = (x.device.type, x.device.index, "conv_wgrad")
    ws, calls = _ws_cache.get(key, (None, 0))
    if ws is None or ws.numel() * 4 < size:
        ws, calls = torch.empty((size + 3) // 4, device=x.device, dtype=torch.float32), 0
        ws[:64].zero_()
    _ws_cache[key] = (ws, calls + 1)
    d_w = torch.empty((c_out, 3, 3, c_in), device=x.device, dtype=torch.float32)
    d_b = torch.empty((c_out,), device=x.device, dtype=torch.float32) if want_bias else None
    hip.check(lib.papr_conv3x3_wgrad(hip.ptr(d_y), hip.ptr(x), B, H, W, c_in, c_out, hip.ptr(d_w), hip.ptr(d_b), d_y_max, x_max, hip.ptr(ws),
                                     calls % 32, hip.stream_ptr()), "papr_conv3x3_wgrad")
    return d_w.permute(0, 3, 1, 2), d_b


from .debug import own_or_raise as _own_or_raise


class _Conv3x3Fn(torch.autograd.Function):
    """relu(conv3x3(x) + b) over an NHWC map with the reference's (C_out, C_in, 3, 3) weight.  Forward and data-gradient on
    papr_conv3x3_fwd, weight gradient on papr_conv3x3_wgrad.  A gradient the own kernels have no form for (data gradient: output channels not a
    multiple of 32; weight gradient: channel counts not multiples of 4, fewer than 32 input channels) raises by name; aten.convolution_backward only
    behind PAPR_DEBUG_TORCH_HEAD=wgrad (papr_amd/debug.py)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        y, ctx.x_max = conv3x3_rows(x, weight.detach(), bias.detach() if bias is not None else None, relu, want_max=True)
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, d_y):
        x, weight, y = ctx.saved_tensors
        d_y = d_y.contiguous()
        if ctx.relu:
            d_y = torch.ops.aten.threshold_backward(d_y, y, 0)
        d_x = d_w = d_b = None
        want_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        shape = "Conv2d(%d, %d, 3x3)" % (weight.shape[1], weight.shape[0])
        # (the kernel's K slabs are 32 channels)
        own_dx = ctx.needs_input_grad[0] and _own_or_raise(weight.shape[0] % 32 == 0, "wgrad", "data gradient of " + shape, "output channels a multiple of 32")
        dy_max = None
        if own_dx:
            d_x, dy_max = conv3x3_rows(d_y, weight, None, False, transposed=True, want_max=True)
        lib_dx = ctx.needs_input_grad[0] and not own_dx
        # (the 32-channel first layer has its own tile shape in the kernel: 128 x 32 instead of 128 x 128)
        own_dw = want_w and _own_or_raise(weight.shape[0] % 4 == 0 and weight.shape[1] % 4 == 0 and weight.shape[1] >= 32, "wgrad", "weight gradient of " + shape,
                                          "channel counts multiples of 4 and at least 32 input channels")
        if own_dw:
            d_w, d_b = conv3x3_wgrad_rows(d_y, x, ctx.needs_input_grad[2], _conv_max_ptr(dy_max, x.device) if dy_max else None,
                                          _conv_max_ptr(ctx.x_max, x.device))
        if (want_w and not own_dw) or lib_dx:           # (PAPR_DEBUG_TORCH_HEAD=wgrad only: _own_or_raise has raised otherwise)
            g_x, g_w, g_b = torch.ops.aten.convolution_backward(d_y.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2), weight, [weight.shape[0]],
                                                                [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [lib_dx, not own_dw, not own_dw])
            if not own_dw:
                d_w, d_b = g_w, g_b
            if lib_dx:
                d_x = g_x.permute(0, 2, 3, 1)
        return d_x, d_w, d_b, None


def _need_hip_rows(t, what):
    if not t.is_cuda:
        raise RuntimeError("papr_amd: %s runs only on a ROCm device (HIP); there is no CPU fallback" % what)
    assert t.dtype == torch.float32 and t.is_contiguous()


class _MaxPool2Fn(torch.autograd.Function):
    """MaxPool2d(2) over an NHWC map (papr_maxpool2_fwd / _bwd; the reference's Down stage, models/unet.py:36-49).  The
    gradient goes to the first maximum of a window in scan order, like torch's."""

    @staticmethod
    def forward(ctx, x):
        _need_hip_rows(x, "the max-pool kernel")
        B, H, W, Cn = x.shape
        out = torch.empty((B, H // 2, W // 2, Cn), device=x.device, dtype=torch.float32)
        which = torch.empty((out.numel() // 4,), device=x.device, dtype=torch.int32) if ctx.needs_input_grad[0] else None
        hip.check(hip.lib().papr_maxpool2_fwd(hip.ptr(x), B, H, W, Cn, hip.ptr(out), hip.ptr(which), hip.stream_ptr()), "papr_maxpool2_fwd")
        ctx.shape = (B, H, W, Cn)
        ctx.save_for_backward(which)
        return out

    @staticmethod
    def backward(ctx, d_out):
        which, = ctx.saved_tensors
        B, H, W, Cn = ctx.shape
        d_out = d_out.contiguous()
        d_in = torch.empty((B, H, W, Cn), device=d_out.device, dtype=torch.float32)
        hip.check(hip.lib().papr_maxpool2_bwd(hip.ptr(d_out), hip.ptr(which), B, H, W, Cn, hip.ptr(d_in), hip.stream_ptr()), "papr_maxpool2_bwd")
        return d_in


class _UpConv2x2Fn(torch.autograd.Function):
    """ConvTranspose2d(kernel 2, stride 2) over an NHWC map with the reference's (C_in, C_out, 2, 2) weight (models/unet.py:62):
    forward, data-gradient and weight-gradient each one launch of the split-f16 kernel of unet.hip (papr_upconv2x2_*)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_hip_rows(x, "the transposed-convolution kernel")
        B, H, W, c_in = x.shape
        c_out = weight.shape[1]
        wm = weight.detach().permute(0, 2, 3, 1).contiguous()        # (no copy for a channels-last parameter)
        out = torch.empty((B, 2 * H, 2 * W, c_out), device=x.device, dtype=torch.float32)
        hip.check(hip.lib().papr_upconv2x2_fwd(hip.ptr(x), B, H, W, c_in, hip.ptr(wm), hip.ptr(bias.detach() if bias is not None else None), c_out,
                                               hip.ptr(out), hip.stream_ptr()), "papr_upconv2x2_fwd")
        ctx.save_for_backward(x, wm)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, wm = ctx.saved_tensors
        B, H, W, c_in = x.shape
        c_out = wm.shape[3]
        d_out = d_out.contiguous()
        lib = hip.lib()
        d_x = d_w = d_b = None
        if ctx.needs_input_grad[0]:
            d_x = torch.empty_like(x)
            hip.check(lib.papr_upconv2x2_dgrad(hip.ptr(d_out), B, H, W, c_in, hip.ptr(wm), c_out, hip.ptr(d_x), hip.stream_ptr()), "papr_upconv2x2_dgrad")
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            d_wm = torch.empty_like(wm)
            d_b = torch.empty((c_out,), device=x.device, dtype=torch.float32) if ctx.needs_input_grad[2] else None
            ws = torch.empty(((lib.papr_upconv2x2_wgrad_workspace_bytes(B, H, W, c_in, c_out) + 3) // 4,), device=x.device, dtype=torch.float32)
            hip.check(lib.papr_upconv2x2_wgrad(hip.ptr(d_out), hip.ptr(x), B, H, W, c_in, c_out, hip.ptr(d_wm), hip.ptr(d_b), hip.ptr(ws), hip.stream_ptr()),
                      "papr_upconv2x2_wgrad")
            d_w = d_wm.permute(0, 3, 1, 2)
        return d_x, d_w, d_b


class _Conv1x1Fn(torch.autograd.Function):
    """The 1x1 output convolution (models/unet.py:86-93) over an NHWC map, (C_out <= 4, C_in, 1, 1) weight: papr_conv1x1_fwd / _bwd."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_hip_rows(x, "the 1x1 convolution kernel")
        B, H, W, c_in = x.shape
        c_out = weight.shape[0]
        w2 = weight.detach().reshape(c_out, c_in).contiguous()
        out = torch.empty((B, H, W, c_out), device=x.device, dtype=torch.float32)
        hip.check(hip.lib().papr_conv1x1_fwd(hip.ptr(x), B * H * W, c_in, hip.ptr(w2), hip.ptr(bias.detach() if bias is not None else None), c_out,
                                             hip.ptr(out), hip.stream_ptr()), "papr_conv1x1_fwd")
        ctx.save_for_backward(x, w2)
        ctx.wshape = weight.shape
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, w2 = ctx.saved_tensors
        B, H, W, c_in = x.shape
        c_out, M = w2.shape[0], B * H * W
        d_out = d_out.contiguous()
        lib = hip.lib()
        d_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        want_w = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        d_w = torch.empty_like(w2) if want_w else None
        d_b = torch.empty((c_out,), device=x.device, dtype=torch.float32) if ctx.needs_input_grad[2] else None
        ws = torch.empty(((lib.papr_conv1x1_bwd_workspace_bytes(M, c_in, c_out) + 3) // 4,), device=x.device, dtype=torch.float32) if want_w else None
        hip.check(lib.papr_conv1x1_bwd(hip.ptr(d_out), hip.ptr(x), M, c_in, hip.ptr(w2), c_out, hip.ptr(d_x), hip.ptr(d_w), hip.ptr(d_b), hip.ptr(ws),
                                       hip.stream_ptr()), "papr_conv1x1_bwd")
        return d_x, (d_w.reshape(ctx.wshape) if want_w else None), d_b


class _SmallUNetFn(torch.autograd.Function):
    """out (B, H, W, n_classes) = SmallUNet(x (B, H, W, c_in)) as ONE library call each way (papr_small_unet_fwd / _bwd; the reference's
    SmallUNet.forward, models/unet.py:206-258): all weight splits and the input maximum in one launch, every other tensor maximum from the kernel
    that produces the tensor, skip concatenations written in place, ReLU masks and the skip tensors' gradient sums inside the kernels at the seams.
    params: conv weight x5, conv bias x5 (inc, down1, down2, up1.conv, up2.conv), up weight x2, up bias x2, out weight, out bias -- the reference's
    parameter tensors as they are."""

    @staticmethod
    def _desc(x, params, one_product):
        B, H, W, c_in = x.shape
        cw, cb, uw, ub, ow, ob = params[0:5], params[5:10], params[10:12], params[12:14], params[14], params[15]
        d = hip.UnetDesc()
        d.B, d.H, d.W, d.c_in, d.n_classes, d.one_product = B, H, W, c_in, ow.shape[0], 1 if one_product else 0
        keep = []
        for i in range(5):
            w = cw[i].detach()
            d.conv_w[i] = w.data_ptr()
            for j, st in enumerate(w.stride()):
                d.conv_w_stride[i][j] = st
            bb = cb[i].detach().contiguous()
            d.conv_b[i] = bb.data_ptr()
            keep += [w, bb]
        wms = []
        for j in range(2):
            wm = uw[j].detach().permute(0, 2, 3, 1).contiguous()        # (no copy for a channels-last parameter)
            bb = ub[j].detach().contiguous()
            d.up_w[j], d.up_b[j] = wm.data_ptr(), bb.data_ptr()
            wms.append(wm)
            keep += [wm, bb]
        w2 = ow.detach().reshape(ow.shape[0], ow.shape[1]).contiguous()
        b2 = ob.detach().contiguous()
        d.out_w, d.out_b = w2.data_ptr(), b2.data_ptr()
        keep += [w2, b2]
        return d, keep, wms

    @staticmethod
    def forward(ctx, x, track, one_product, *params):
        _need_hip_rows(x, "the U-Net head")
        lib = hip.lib()
        B, H, W, c_in = x.shape
        keep_state = bool(track and (ctx.needs_input_grad[0] or any(ctx.needs_input_grad[3:])))
        ctx.one_product = one_product
        d, alive, _ = _SmallUNetFn._desc(x, params, one_product)
        state = torch.empty(lib.papr_small_unet_state_bytes(B, H, W, c_in, 1 if keep_state else 0), device=x.device, dtype=torch.uint8)
        out = torch.empty((B, H, W, d.n_classes), device=x.device, dtype=torch.float32)
        hip.check(lib.papr_small_unet_fwd(C.byref(d), hip.ptr(x), hip.ptr(out), hip.ptr(state), 1 if keep_state else 0, hip.stream_ptr()), "papr_small_unet_fwd")
        if keep_state:
            ctx.save_for_backward(x, state, *params)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, state = ctx.saved_tensors[:2]
        params = ctx.saved_tensors[2:]
        lib = hip.lib()
        B, H, W, c_in = x.shape
        dev = x.device
        d, alive, wms = _SmallUNetFn._desc(x, params, ctx.one_product)
        g = hip.UnetGrads()
        E = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
        d_cw = [E(params[i].shape[0], 3, 3, params[i].shape[1]) for i in range(5)]
        d_cb = [E(params[i].shape[0]) for i in range(5)]
        d_uw = [torch.empty_like(wm) for wm in wms]
        d_ub = [E(wm.shape[3]) for wm in wms]
        d_ow, d_ob = E(d.n_classes, params[14].shape[1]), E(d.n_classes)
        for i in range(5):
            g.conv_w[i], g.conv_b[i] = d_cw[i].data_ptr(), d_cb[i].data_ptr()
        for j in range(2):
            g.up_w[j], g.up_b[j] = d_uw[j].data_ptr(), d_ub[j].data_ptr()
        g.out_w, g.out_b = d_ow.data_ptr(), d_ob.data_ptr()
        d_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ws = torch.empty(lib.papr_small_unet_bwd_workspace_bytes(B, H, W, c_in, d.n_classes), device=dev, dtype=torch.uint8)
        hip.check(lib.papr_small_unet_bwd(C.byref(d), hip.ptr(x), hip.ptr(d_out.contiguous()), hip.ptr(state), hip.ptr(d_x), C.byref(g), hip.ptr(ws), hip.stream_ptr()),
                  "papr_small_unet_bwd")
        grads = [t.permute(0, 3, 1, 2) for t in d_cw] + d_cb + [t.permute(0, 3, 1, 2) for t in d_uw] + d_ub + [d_ow.reshape(params[14].shape), d_ob]
        return (d_x, None, None) + tuple(grads)


def small_unet_rows(x, net, one_product=False):
    """x (B, H, W, c_in) contiguous -> (B, H, W, n_classes): `net` = a papr_amd.unet.SmallUNet (the reference's module attribute names).  one_product:
    one f16 product per fp32 product in the 3x3 and transposed convolutions (the arithmetic of the reference's fp16 autocast, `use_amp: true`)."""
    c = lambda m: m.double_conv[0]
    convs = [c(net.inc), c(net.down1.maxpool_conv[1]), c(net.down2.maxpool_conv[1]), c(net.up1.conv), c(net.up2.conv)]
    ups = [net.up1.up, net.up2.up]
    params = [m.weight for m in convs] + [m.bias for m in convs] + [m.weight for m in ups] + [m.bias for m in ups] + [net.outc.conv.weight, net.outc.conv.bias]
    return _SmallUNetFn.apply(x, torch.is_grad_enabled(), bool(one_product), *params)


class _CompositeFn(torch.autograd.Function):
    """rgb = fg * (1 - a) + bkg * a (normalize) or fg + bkg * a, a = the background token's attention: the last line of the
    reference's forward (models/model.py:536-545) and its autograd as one launch forward, two backward (papr_composite_fwd / _bwd)
    instead of ~20 elementwise / reduction launches."""

    @staticmethod
    def forward(ctx, fg, attn, bkg, normalize):
        _need_hip_rows(fg, "the compositing kernel")
        Cn = fg.shape[-1]
        R = fg.numel() // Cn
        attn = attn.contiguous()
        bkg_c = bkg.detach().reshape(-1).contiguous()
        rgb = torch.empty_like(fg)
        hip.check(hip.lib().papr_composite_fwd(hip.ptr(fg), hip.ptr(attn), attn.shape[-1], attn.shape[-1] - 1, hip.ptr(bkg_c), R, Cn, 1 if normalize else 0,
                                               hip.ptr(rgb), hip.stream_ptr()), "papr_composite_fwd")
        ctx.save_for_backward(fg, attn, bkg_c)
        ctx.normalize, ctx.bkg_shape = normalize, bkg.shape
        return rgb

    @staticmethod
    def backward(ctx, d_rgb):
        fg, attn, bkg_c = ctx.saved_tensors
        Cn = fg.shape[-1]
        R = fg.numel() // Cn
        d_rgb = d_rgb.contiguous()
        lib = hip.lib()
        d_fg = torch.empty_like(fg) if ctx.needs_input_grad[0] else None
        d_attn = torch.empty_like(attn) if ctx.needs_input_grad[1] else None
        d_bkg = torch.empty_like(bkg_c) if ctx.needs_input_grad[2] else None
        ws = torch.empty(((lib.papr_composite_bwd_workspace_bytes(R) + 3) // 4,), device=fg.device, dtype=torch.float32) if d_bkg is not None else None
        hip.check(lib.papr_composite_bwd(hip.ptr(d_rgb), hip.ptr(fg), hip.ptr(attn), attn.shape[-1], attn.shape[-1] - 1, hip.ptr(bkg_c), R, Cn,
                                         1 if ctx.normalize else 0, hip.ptr(d_fg), hip.ptr(d_attn), hip.ptr(d_bkg), hip.ptr(ws), hip.stream_ptr()),
                  "papr_composite_bwd")
        return d_fg, d_attn, (d_bkg.reshape(ctx.bkg_shape) if d_bkg is not None else None), None


class _LnFoldFn(torch.autograd.Function):
    """(W * a_2 zero-padded to ld_eff columns, c + W b_2): the LayerNorm affine in front of a Linear layer folded into it, one
    launch each way (in torch ops: two products, a row sum and a pad forward, eight small kernels backward, two of them
    16 us column sums)."""

    @staticmethod
    def forward(ctx, w, c, a2, b2, ld_eff):
        w, c, a2, b2 = w.contiguous(), c.contiguous(), a2.contiguous(), b2.contiguous()
        n_out, n_in = w.shape
        eff_w = torch.empty((n_out, ld_eff), device=w.device, dtype=torch.float32)
        eff_b = torch.empty((n_out,), device=w.device, dtype=torch.float32)
        hip.check(hip.lib().papr_ln_fold_fwd(hip.ptr(w), n_out, n_in, n_in, hip.ptr(c), hip.ptr(a2), hip.ptr(b2), hip.ptr(eff_w), ld_eff,
                                              hip.ptr(eff_b), hip.stream_ptr()), "papr_ln_fold_fwd")
        ctx.save_for_backward(w, a2, b2)
        ctx.ld_eff = ld_eff
        return eff_w, eff_b

    @staticmethod
    def backward(ctx, d_eff_w, d_eff_b):
        w, a2, b2 = ctx.saved_tensors
        n_out, n_in = w.shape
        d_eff_w, d_eff_b = d_eff_w.contiguous(), d_eff_b.contiguous()
        d_w, d_a2, d_b2 = torch.empty_like(w), torch.empty_like(a2), torch.empty_like(b2)
        hip.check(hip.lib().papr_ln_fold_bwd(hip.ptr(w), n_out, n_in, n_in, hip.ptr(a2), hip.ptr(b2), hip.ptr(d_eff_w), ctx.ld_eff,
                                              hip.ptr(d_eff_b), hip.ptr(d_w), hip.ptr(d_a2), hip.ptr(d_b2), hip.stream_ptr()), "papr_ln_fold_bwd")
        return d_w, d_eff_b, d_a2, d_b2, None


class _LnFoldBatchFn(torch.autograd.Function):
    """_LnFoldFn for several layers at once: one launch each way (papr_ln_fold_fwd_batch / _bwd_batch).  Arguments: the tuple of the layers' ld_eff, then
    (w, c, a2, b2) per layer; returns (eff_w, eff_b) per layer, flat."""

    @staticmethod
    def _jobs(n):
        return (hip.LnFoldJob * n)()

    @staticmethod
    def forward(ctx, ld_effs, *t):
        n = len(ld_effs)
        t = [x.contiguous() for x in t]
        jobs, outs = _LnFoldBatchFn._jobs(n), []
        for i in range(n):
            w, c, a2, b2 = t[4 * i:4 * i + 4]
            n_out, n_in = w.shape
            eff_w = torch.empty((n_out, ld_effs[i]), device=w.device, dtype=torch.float32)
            eff_b = torch.empty((n_out,), device=w.device, dtype=torch.float32)
            j = jobs[i]
            j.w, j.n_out, j.n_in, j.ldw, j.c, j.a2, j.b2 = w.data_ptr(), n_out, n_in, n_in, c.data_ptr(), a2.data_ptr(), b2.data_ptr()
            j.eff_w, j.ld_eff, j.eff_b = eff_w.data_ptr(), ld_effs[i], eff_b.data_ptr()
            outs += [eff_w, eff_b]
        hip.check(hip.lib().papr_ln_fold_fwd_batch(jobs, n, hip.stream_ptr()), "papr_ln_fold_fwd_batch")
        ctx.save_for_backward(*[x for i in range(n) for x in (t[4 * i], t[4 * i + 2], t[4 * i + 3])])
        ctx.ld_effs = ld_effs
        return tuple(outs)

    @staticmethod
    def backward(ctx, *d):
        n = len(ctx.ld_effs)
        sv = ctx.saved_tensors
        jobs, grads, alive = _LnFoldBatchFn._jobs(n), [], []
        for i in range(n):
            w, a2, b2 = sv[3 * i:3 * i + 3]
            n_out, n_in = w.shape
            d_eff_w = d[2 * i].contiguous() if d[2 * i] is not None else torch.zeros((n_out, ctx.ld_effs[i]), device=w.device, dtype=torch.float32)
            d_eff_b = d[2 * i + 1].contiguous() if d[2 * i + 1] is not None else torch.zeros((n_out,), device=w.device, dtype=torch.float32)
            d_w, d_a2, d_b2 = torch.empty_like(w), torch.empty_like(a2), torch.empty_like(b2)
            j = jobs[i]
            j.w, j.n_out, j.n_in, j.ldw, j.a2, j.b2 = w.data_ptr(), n_out, n_in, n_in, a2.data_ptr(), b2.data_ptr()
            j.d_eff_w, j.ld_eff, j.d_eff_b, j.d_w, j.d_a2, j.d_b2 = d_eff_w.data_ptr(), ctx.ld_effs[i], d_eff_b.data_ptr(), d_w.data_ptr(), d_a2.data_ptr(), d_b2.data_ptr()
            grads += [d_w, d_eff_b, d_a2, d_b2]
            alive += [d_eff_w, d_eff_b]
        hip.check(hip.lib().papr_ln_fold_bwd_batch(jobs, n, hip.stream_ptr()), "papr_ln_fold_bwd_batch")
        return (None,) + tuple(grads)


def ln_fold_job(spec, weights, biases, ln_in):
    """(w, c, a2, b2, ld_eff) if the first layer of `spec` folds the LayerNorm affine `ln_in` on the device kernel (what prepare_mlp_weights would
    hand to _LnFoldFn), else None."""
    L, w = spec.layers[0], weights[0]
    if L["skip"] or not w.is_cuda or L["n_in"] > 1024:
        return None
    if ln_in is None:
        # no LayerNorm in front, but the first layer's rows want zero padding to the kernel's row length (F.pad: a fill and a copy forward, a slice
        # copy backward): the same kernel with a_2 = 1, b_2 = 0 does it inside the launch the other folds already pay for
        if L["n_in"] == w.shape[1] or L["n_out_pad"] != L["n_out"]:
            return None
        return (w, biases[0], _const(w.device, w.shape[1], 1.0), _zeros(w.device, (w.shape[1],)), L["n_in"])
    return (w, biases[0], ln_in[0], ln_in[1], L["n_in"])


def ln_fold_batch(folds):
    """[(w, c, a2, b2, ld_eff), ...] -> [(eff_w, eff_b), ...] in one launch each way."""
    out = _LnFoldBatchFn.apply(tuple(f[4] for f in folds), *[x for f in folds for x in f[:4]])
    return [(out[2 * i], out[2 * i + 1]) for i in range(len(folds))]


def prepare_mlp_weights(spec, weights, biases, ln_in=None, prefolded=None):
    """Reference-shaped Linear parameters -> effective, padded weights for the kernels (differentiable).

    ln_in = (a_2, b_2) of the input LayerNorm: the kernels only standardise rows, the affine part is
    folded here:  W (a * xh + b) + c = (W * a) xh + (W b + c).  prefolded = (eff_w, eff_b) of the first layer when the caller has folded it already
    (ln_fold_batch: all folds of a model in one launch).
    """
    eff_w, eff_b = [], []
    for i, (w, b) in enumerate(zip(weights, biases)):
        L = spec.layers[i]
        # (no slice when the layer has no skip block: its backward would zero-fill and copy a full-size gradient)
        main = w[:, :L["raw_in"]] if L["skip"] else w
        extra = w[:, L["raw_in"]:] if L["skip"] else None
        if prefolded is not None and i == 0:
            main, b = prefolded
        elif ln_in is not None and i == 0 and extra is None and w.is_cuda and L["n_in"] <= 1024:
            main, b = _LnFoldFn.apply(main, b, ln_in[0], ln_in[1], L["n_in"])      # (already padded to the kernel's row length)
        elif ln_in is not None:
            a, sh = ln_in
            if i == 0:
                b = b + (main * sh).sum(1)      # (not `main @ sh`: rocBLAS' gemv takes 58 us for a 256 x 117 matrix)
                main = main * a
            if extra is not None:
                b = b + (extra * sh).sum(1)
                extra = extra * a
        # (F.pad with nothing to pad still copies, forward and backward: 38 launches per step for the default model)
        if L["n_in"] != main.shape[1]:
            main = F.pad(main, (0, L["n_in"] - main.shape[1]))
        if extra is not None:
            if spec.ld_in != extra.shape[1]:
                extra = F.pad(extra, (0, spec.ld_in - extra.shape[1]))
            main = torch.cat([main, extra], dim=1)
        if L["n_out_pad"] != L["n_out"]:
            main = F.pad(main, (0, 0, 0, L["n_out_pad"] - L["n_out"]))
            b = F.pad(b, (0, L["n_out_pad"] - L["n_out"]))
        eff_w.append(main.contiguous())
        eff_b.append(b.contiguous())
    return eff_w, eff_b


def _layer_table(spec, ws, bs, wts=None):
    tab = (hip.Layer * spec.n_layer)()
    for i, L in enumerate(spec.layers):
        t = tab[i]
        t.weight = ws[i].data_ptr()
        t.weight_t = wts[i].data_ptr() if wts is not None else None
        t.bias = bs[i].data_ptr()
        t.n_in, t.n_out = L["n_in"], L["n_out_pad"]
        t.ldw = ws[i].shape[1]
        t.ldwt = wts[i].shape[1] if wts is not None else 0
        t.n_skip = spec.ld_in if L["skip"] else 0
        t.skip_col = L["n_in"] if L["skip"] else 0
        t.act = L["act"]
    return tab


class LayerOutputs(list):
    """The saved activations of one chain plus the per-row input maxima papr_mlp_fwd leaves for papr_mlp_bwd."""
    row_absmax = None
    norm_stats = None
    norm_mean = None       # (raw_rows) the last output holds the UN-standardised rows, this their means
    in_stats = None
    dots = None
    wkT = None             # (the w_q / w_k run: W_k^T as the forward pass laid it out)


def mlp_forward(spec, ws, bs, x, M, keep=True, out_norm=None, in_norm=None, dot_rows=None, rows_per_dot=1, raw_rows=False, leave_input=False):
    """Run the chain; returns the list of layer outputs (all kept, or two ping-pong buffers).
    out_norm = (width, eps): the last output comes back row-standardised (LayerNorm core), outs.norm_stats holds
    the (M, 2) statistics papr_rownorm_bwd needs.  in_norm = (width, eps): x is standardised first (outs.in_stats; x itself
    is overwritten, except in inference inside a fused run, where nobody reads it again).
    dot_rows (with out_norm): outs.dots (M,) = standardised row m . dot_rows[m // rows_per_dot]; without `keep` the last output
    itself is then undefined (a fused run does not write it).  raw_rows (with dot_rows and keep; fused runs only -- the library refuses
    otherwise): the last output stays UN-standardised, outs.norm_mean (M,) holds the rows' means (papr_row_norm.raw_mean)."""
    dev = x.device
    outs = LayerOutputs()
    if keep:
        outs.extend(torch.empty((M, ld), device=dev, dtype=torch.float32) for ld in spec.ld_out)
        outs.row_absmax = torch.empty(hip.lib().papr_mlp_saved_floats(spec.n_layer, M), device=dev, dtype=torch.float32)
    else:
        pool = [torch.empty(M * max(spec.ld_out), device=dev, dtype=torch.float32) for _ in range(2)]
        outs.extend(pool[i & 1][:M * ld].view(M, ld) for i, ld in enumerate(spec.ld_out))
    tab = _layer_table(spec, ws, bs)
    norm = inorm = None
    if in_norm is not None:
        # (width, eps) or (width, eps, stats, mean): the statistics GIVEN by the kernel that wrote x (papr_build_features_fwd) -- the call applies them
        given = in_norm[2:] if len(in_norm) > 2 else None
        outs.in_stats = given[0] if given else torch.empty((M, 2), device=dev, dtype=torch.float32)
        rn_in = hip.RowNorm(in_norm[1], in_norm[0], outs.in_stats.data_ptr())
        if given:
            rn_in.given_mean = given[1].data_ptr()
        rn_in.leave_input = 1 if leave_input else 0      # (the caller's backward pass of the norm reads neither x nor the standardised rows)
        inorm = C.byref(rn_in)
    if out_norm is not None:
        outs.norm_stats = torch.empty((M, 2), device=dev, dtype=torch.float32)
        rn = hip.RowNorm(out_norm[1], out_norm[0], outs.norm_stats.data_ptr())
        if dot_rows is not None:
            outs.dots = torch.empty(M, device=dev, dtype=torch.float32)
            rn.dot_rows, rn.ld_dot, rn.rows_per_dot, rn.dots = dot_rows.data_ptr(), dot_rows.stride(0), rows_per_dot, outs.dots.data_ptr()
            if raw_rows and keep:
                outs.norm_mean = torch.empty(M, device=dev, dtype=torch.float32)
                rn.raw_mean = outs.norm_mean.data_ptr()
        norm = C.byref(rn)
    hip.check(hip.lib().papr_mlp_fwd(tab, spec.n_layer, hip.ptr(x), x.shape[1], M, hip.ptr_array(outs),
                                     hip.i32_array(spec.ld_out), hip.ptr(outs.row_absmax), inorm, norm,
                                     hip.ptr(_workspace(dev, "fwd", M)), mlp_mode(spec.one_product, getattr(spec, "fp32_rows_inside", False)), hip.stream_ptr()), "papr_mlp_fwd")
    return outs


_ws_cache = {}
_SCORES_IN_RUN = os.environ.get("PAPR_SCORES_IN_RUN", "1") != "0"
_QK_CHAIN = int(os.environ.get("PAPR_QK_CHAIN", "2"))           # A/B: 0 = w_q and the w_k fold as two single-layer calls each way; 1 = one two-layer run forward; 2 = backward as well
_LN_IN_FEATURES = os.environ.get("PAPR_LN_IN_FEATURES", "1") != "0"     # A/B: the key in-norm's backward pass inside papr_build_features_bwd_pairs (0: papr_rownorm_bwd)
_RAW_KEYS = os.environ.get("PAPR_RAW_KEYS", "1") != "0"        # A/B: training keeps the key embedding un-standardised (papr_row_norm.raw_mean), tail_bwd standardises on the fly
_QK_BIAS_KERNEL = os.environ.get("PAPR_QK_BIAS_KERNEL", "1") != "0"   # A/B: the score bias' rank-one gradient terms on papr_qk_bias_bwd (0: torch ops)
_KEY_STATS = os.environ.get("PAPR_KEY_STATS", "1") != "0"      # A/B: the key rows' LayerNorm statistics from papr_build_features_fwd (default) or from the fused run


def _workspace(dev, kind, M):
    """Grow-only scratch buffer per (device, kind); calls on one stream run in order, so it is shared."""
    size = hip.lib().papr_mlp_fwd_workspace_bytes(M) if kind == "fwd" else hip.lib().papr_mlp_bwd_workspace_bytes(M)
    key = (dev.type, dev.index, kind)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() * 4 < size:
        buf = _ws_cache[key] = torch.empty((size + 3) // 4, device=dev, dtype=torch.float32)
    return buf


_TAIL_F16 = os.environ.get("PAPR_TAIL_F16", "1") != "0"        # A/B: one-product mode -- papr_attn_tail_bwd writes the key / value gradient rows as the data-gradient run stages them (papr_f16_rows)


TAIL_F16_ROWS_USED = 0       # (tests: counts the row sets handed over that way)


class F16Rows:
    """Device buffers of one papr_f16_rows (include/papr_hip.h) and the struct that points at them."""

    def __init__(self, dev, M, width, split=False):
        ld = (width + 31) // 32 * 32
        self.hi = torch.empty((M, ld), device=dev, dtype=torch.float16)
        self.lo = torch.empty((M, ld), device=dev, dtype=torch.float16) if split else None      # (the parity runs' form: hi + lo)
        self.tables = torch.empty((3, M), device=dev, dtype=torch.float32)       # 1 / scale, scale, max |row|
        self.c = hip.F16Rows(self.hi.data_ptr(), self.tables[0].data_ptr(), self.tables[1].data_ptr(), self.tables[2].data_ptr(), ld,
                             self.lo.data_ptr() if split else None)

    def ref(self):
        return C.byref(self.c)

    def float(self):
        """The rows as fp32 (tests)."""
        return (self.hi.float() + (self.lo.float() if self.lo is not None else 0.0)) * self.tables[0][:, None]


def mlp_backward_takes_f16(spec, ws, bs, need_dx):
    """Whether mlp_backward(..., d_out=F16Rows) is accepted for this chain (papr_mlp_bwd_takes_f16_rows): 0 no, 1 the one-product form,
    2 the parity form (F16Rows(..., split=True))."""
    if not _TAIL_F16:
        return 0
    return int(hip.lib().papr_mlp_bwd_takes_f16_rows(_layer_table(spec, ws, bs), spec.n_layer, hip.i32_array(spec.ld_out), 1 if need_dx else 0,
                                                      mlp_mode(spec.one_product, getattr(spec, "fp32_rows_inside", False))))


def mlp_backward(spec, ws, bs, x, M, outs, d_out, scratch, need_dx):
    """Returns (d_ws, d_bs, d_x or None).  d_out is consumed; an F16Rows instead of a tensor: see mlp_backward_takes_f16."""
    dev = x.device
    d_out_f16 = None
    if isinstance(d_out, F16Rows):
        d_out, d_out_f16 = None, d_out.ref()
    tab = _layer_table(spec, ws, bs)
    if hip.lib().papr_mlp_bwd_needs_weight_t(tab, spec.n_layer, 1 if need_dx else 0, mlp_mode(spec.one_product, getattr(spec, "fp32_rows_inside", False))):     # (fused runs read W^T in place)
        wts = [w.t().contiguous() for w in ws]
        tab = _layer_table(spec, ws, bs, wts)
    d_ws = [torch.empty_like(w) for w in ws]
    d_bs = [torch.empty_like(b) for b in bs]
    d_x = torch.empty_like(x) if need_dx else None
    hip.check(hip.lib().papr_mlp_bwd(tab, spec.n_layer, hip.ptr(x), x.shape[1], M, hip.ptr_array(outs),
                                     hip.i32_array(spec.ld_out), hip.ptr(getattr(outs, "row_absmax", None)),
                                     hip.ptr(d_out), d_out_f16, hip.ptr(scratch[0]), hip.ptr(scratch[1]),
                                     scratch[0].shape[1], hip.ptr_array(d_ws), hip.ptr_array(d_bs), hip.ptr(d_x),
                                     hip.ptr(_workspace(dev, "bwd", M)), mlp_mode(spec.one_product, getattr(spec, "fp32_rows_inside", False)), hip.stream_ptr()), "papr_mlp_bwd")
    return d_ws, d_bs, d_x


class _MlpRowsFn(torch.autograd.Function):
    """y (M, ld_out) = MLP(x (M, ld_in)) on the library's GEMM kernels, for modules that are just an MLP over rows
    (the per-pixel MLPGenerator render head).  Weights arrive prepared (prepare_mlp_weights), so their gradients flow on
    through ordinary autograd."""

    @staticmethod
    def forward(ctx, spec, track, x, n, *wb):
        ws, bs = list(wb[:n]), list(wb[n:])
        M = x.shape[0]
        keep = track and (ctx.needs_input_grad[2] or any(ctx.needs_input_grad[4:]))
        outs = mlp_forward(spec, ws, bs, x, M, keep)
        ctx.spec, ctx.n = spec, n
        if keep:
            ctx.saved = (x, outs, ws, bs)
        return outs[-1] if keep else outs[-1].clone()

    @staticmethod
    def backward(ctx, d_y):
        spec, n = ctx.spec, ctx.n
        x, outs, ws, bs = ctx.saved
        M = x.shape[0]
        wmax = max([spec.width, spec.ld_in] + list(spec.ld_out))
        scratch = [torch.empty((M, wmax), device=x.device, dtype=torch.float32) for _ in range(2)]
        d_ws, d_bs, d_x = mlp_backward(spec, ws, bs, x, M, outs, d_y.contiguous().clone(), scratch, ctx.needs_input_grad[2])
        ctx.saved = None
        return (None, None, d_x, None) + tuple(d_ws) + tuple(d_bs)


def mlp_rows(spec, x, weights, biases):
    """Differentiable MLP over the rows of x (M, d_in) with reference-shaped Linear parameters; returns (M, d_out)."""
    if not x.is_cuda:
        raise RuntimeError("papr_amd: the MLP kernels run only on a ROCm device (HIP); there is no CPU fallback")
    ew, eb = prepare_mlp_weights(spec, weights, biases)
    xp = x if x.shape[1] == spec.ld_in else F.pad(x, (0, spec.ld_in - x.shape[1]))
    y = _MlpRowsFn.apply(spec, torch.is_grad_enabled(), xp.contiguous(), len(ew), *ew, *eb)
    return y[:, :spec.d_out]


def linear_rows(x, w):
    """out = x @ w.T on the MFMA GEMM (no bias, no activation).  x (M, ldx), w (n_out, ldx) with n_out % 4 == 0.
    Row results do not depend on M (unlike a library GEMM, whose tiling follows the problem size), which keeps
    chunked evaluation bit-identical to whole-image evaluation."""
    M, n_out = x.shape[0], w.shape[0]
    out = torch.empty((M, n_out), device=x.device, dtype=torch.float32)
    tab = (hip.Layer * 1)()
    t = tab[0]
    t.weight, t.weight_t, t.bias = w.data_ptr(), None, None
    t.n_in, t.n_out, t.ldw, t.ldwt, t.n_skip, t.skip_col, t.act = x.shape[1], n_out, w.shape[1], 0, 0, 0, 0
    hip.check(hip.lib().papr_mlp_fwd(tab, 1, hip.ptr(x), x.shape[1], M, hip.ptr_array([out]), hip.i32_array([n_out]),
                                     None, None, None, hip.ptr(_workspace(x.device, "fwd", M)), mlp_mode(False), hip.stream_ptr()), "papr_mlp_fwd")
    return out


def rownorm_(x, width, eps):
    """In-place row standardisation; returns stats (rows,2)."""
    stats = torch.empty((x.shape[0], 2), device=x.device, dtype=torch.float32)
    hip.check(hip.lib().papr_rownorm_fwd(hip.ptr(x), x.shape[0], width, x.shape[1], eps, hip.ptr(x), hip.ptr(stats),
                                         hip.stream_ptr()), "papr_rownorm_fwd")
    return stats


def rownorm_bwd_(dy, y, stats, width, eps):
    """In-place: dy <- gradient w.r.t. the un-normalised rows."""
    hip.check(hip.lib().papr_rownorm_bwd(hip.ptr(dy), hip.ptr(y), hip.ptr(stats), y.shape[0], width, y.shape[1], eps,
                                         hip.ptr(dy), hip.stream_ptr()), "papr_rownorm_bwd")
    return dy


def points_knn(points, k, query_idx=None, want_dist=True):
    """points (P,3) float32 device tensor -> nn_idx (Q,k) int32 [, nn_dist (Q,k) float64] of papr_points_knn (Q = P without query_idx)."""
    P, dev = points.shape[0], points.device
    Q = P if query_idx is None else query_idx.numel()
    nn_idx = torch.empty((Q, k), device=dev, dtype=torch.int32)
    nn_dist = torch.empty((Q, k), device=dev, dtype=torch.float64) if want_dist else None
    hip.check(hip.lib().papr_points_knn(hip.ptr(points), P, None if query_idx is None else hip.ptr(query_idx), Q, k, hip.ptr(nn_idx),
                                        None if nn_dist is None else hip.ptr(nn_dist), hip.stream_ptr()), "papr_points_knn")
    return (nn_idx, nn_dist) if want_dist else nn_idx


def group_pairs(flat_idx, P, run=1):
    """(order int64, sorted_pts int32, seg int64[P+1]) of papr_group_pairs: the pairs grouped by selected point.  run: every aligned run of
    `run` consecutive entries holds distinct points (a ray's k neighbours); 1 = no such promise."""
    M, dev = flat_idx.numel(), flat_idx.device
    lib = hip.lib()
    order = torch.empty(M, device=dev, dtype=torch.int64)
    sorted_pts = torch.empty(M, device=dev, dtype=torch.int32)
    seg = torch.empty(P + 1, device=dev, dtype=torch.int64)
    nb = lib.papr_group_pairs_workspace_bytes(M, P)
    ws = torch.empty(nb, device=dev, dtype=torch.uint8)
    hip.check(lib.papr_group_pairs(hip.ptr(flat_idx), M, P, int(run), hip.ptr(order), hip.ptr(sorted_pts), hip.ptr(seg), hip.ptr(ws), nb, hip.stream_ptr()),
              "papr_group_pairs")
    return order, sorted_pts, seg


def ray_knn(points, rays_o, rays_d, rays_per_image, k, eps, want_dist=False):
    """points (P,3), rays_o (N,3), rays_d (R,3) -> idx (R,k) int32 [, dist (R,k)]  (no gradient)."""
    R = rays_d.shape[0]
    dev = points.device
    idx = torch.empty((R, k), device=dev, dtype=torch.int32)
    dist = torch.empty((R, k), device=dev, dtype=torch.float32) if want_dist else None
    ws = torch.empty(hip.lib().papr_ray_knn_workspace_bytes(R, points.shape[0]) // 4, device=dev, dtype=torch.float32)
    hip.check(hip.lib().papr_ray_knn(hip.ptr(points), points.shape[0], hip.ptr(rays_o), hip.ptr(rays_d), R, rays_per_image,
                                     k, eps, hip.ptr(idx), hip.ptr(dist), hip.ptr(ws), hip.stream_ptr()), "papr_ray_knn")
    return (idx, dist) if want_dist else idx


class RenderPath:
    """Static plan of the per-ray path for one configuration."""

    def __init__(self, cfg):
        a = cfg["models"]["attn"]
        e = a["embed"]
        pf = cfg["geoms"]["point_feats"]
        if a["k_type"] != 1:
            raise ValueError("Invalid key type")
        if a["q_type"] != 1:
            raise ValueError("Invalid query type")
        if a["v_type"] != 1:
            raise ValueError("Invalid value type")
        if e["embed_type"] not in (1, 2):
            raise ValueError("Unknown embedding type: {}".format(e["embed_type"]))
        if pf["use_inq"]:
            raise NotImplementedError("papr_amd: geoms.point_feats.use_inq is not supported (it does not broadcast in the reference either)")
        self.eps = float(cfg["eps"])
        self.k_cfg = int(cfg["geoms"]["points"]["select_k"])
        self.feat_dim = int(pf["dim"])
        self.use_feats = bool(pf["use_ink"] or pf["use_inv"])
        self.d_model = int(a["d_model"])
        self.score_act = a["score_act"].lower()
        if self.score_act not in hip.ACT:
            raise NotImplementedError("papr_amd: score_act '%s' has no HIP implementation" % self.score_act)
        self.bkg_score = float(cfg["geoms"]["background"]["constant"])
        self.normalize = bool(cfg["models"]["normalize_topk_attn"])
        with_self = 1 if e["embed_type"] == 1 else 0
        w = lambda Ls: sum(3 * (with_self + 2 * L) for L in Ls)
        self.key_w = w(e["k_L"]) + (self.feat_dim if pf["use_ink"] else 0)
        self.qry_w = w(e["q_L"])
        self.val_w = w(e["v_L"]) + (self.feat_dim if pf["use_inv"] else 0)
        self.key = MlpSpec("key", self.key_w, e["key"])
        self.qry = MlpSpec("query", self.qry_w, e["query"])
        self.val = MlpSpec("value", self.val_w, e["value"])
        # each FeedForward has its own `norm` (models/attn.py:100-107): LayerNorm in front of and behind the key / query / value MLP independently.
        # Key and query: the out-norm's affine folds into w_k / w_q.  Value: nothing follows its out-norm, so the affine is applied by the host
        # glue (v_norm below; no shipped scene file sets value.norm)
        self.k_norm = self.key.norm == "layernorm"
        self.q_norm = self.qry.norm == "layernorm"
        self.v_norm = self.val.norm == "layernorm"
        self.C = self.val.d_out
        # single-layer "MLPs" for w_k / w_q so that they share the GEMM drivers
        one = lambda name, d_in: MlpSpec(name, d_in, dict(n_ff_layer=1, d_ff=self.d_model, d_ff_out=self.d_model, norm="none",
                                                          ff_act="none", ff_last_act="none"))
        if self.d_model % 4 or self.d_model > 1024:
            raise NotImplementedError("papr_amd: d_model must be a multiple of 4 and <= 1024")
        self.wk = one("w_k", self.key.d_out)
        self.wq = one("w_q", self.qry.d_out)
        # g = q' W_k seen as a Linear layer (weight W_k^T): its backward runs on the library's own GEMMs
        self.wk_fold = MlpSpec("w_k_fold", self.d_model, dict(n_ff_layer=1, d_ff=self.key.d_out, d_ff_out=self.key.d_out, norm="none",
                                                              ff_act="none", ff_last_act="none"))
        # q' = W_q Q + b_q and g = W_k^T q' as ONE two-layer run of R rows (before: two single-layer launches of ~45 us each way, each with its own
        # row-maximum and weight-split launch); half_layers spelled out so that the middle width is d_model whatever key.d_out is
        self.wqk = None
        if self.d_model % 32 == 0 and self.key.d_out % 32 == 0 and self.d_model <= 256 and self.key.d_out <= 256:
            self.wqk = MlpSpec("w_q_k", self.qry.d_out, dict(n_ff_layer=2, d_ff=self.d_model, d_ff_out=self.key.d_out, norm="none",
                                                            ff_act="none", ff_last_act="none"))
            self.wqk.fp32_rows_inside = True        # (q', the run's INNER row, is read by the host side: c0 = q'.b_k and its gradients -- in the
                                                    # process-wide one-product mode a run's inner rows are f16 otherwise)
        d = hip.FeatureDesc()
        d.feat_dim = self.feat_dim
        d.L_key = (C.c_int32 * 3)(*e["k_L"])
        d.L_qry = e["q_L"][0]
        d.L_val = (C.c_int32 * 2)(*e["v_L"])
        d.with_self = with_self
        d.key_has_feats = int(bool(pf["use_ink"]))
        d.val_has_feats = int(bool(pf["use_inv"]))
        d.pe_factor, d.pe_mult, d.eps = float(e["pe_factor"]), float(e["pe_mult_factor"]), self.eps
        d.ld_key, d.ld_qry, d.ld_val = self.key.ld_in, self.qry.ld_in, self.val.ld_in
        self.fdesc = d
        # `use_amp: true` (the reference's shipped default: its attention block then runs under fp16 autocast, models/attn.py:248):
        # the embedding MLPs multiply ONE f16 product per fp32 product and keep f16 rows for their weight gradients (the library's
        # h1 arithmetic; tolerance in tests/test_hip_h1.py).  PAPR_AMP_MLP=fp32 keeps the parity arithmetic under use_amp.
        self.amp_mlp = bool(cfg.get("use_amp", False)) and os.environ.get("PAPR_AMP_MLP", "h1") == "h1"
        only = os.environ.get("PAPR_AMP_MLP_ONLY", "key,query,value").split(",")      # (A/B of scripts/probes: which of the three take the one-product arithmetic)
        for spec, name in ((self.key, "key"), (self.qry, "query"), (self.val, "value")):
            spec.one_product = self.amp_mlp and name in only

    # -------------------------------------------------------------------------------------------
    def feature_desc(self, k):
        d = hip.FeatureDesc()
        C.memmove(C.byref(d), C.byref(self.fdesc), C.sizeof(d))
        d.k = k
        return d

    def tail_desc(self, k):
        t = hip.TailDesc()
        # kp operand = the key embedding rows themselves, qp operand = W_k^T (W_q Q + b_q): see render path below
        t.k, t.d_model, t.C = k, self.key.d_out, self.C
        t.scale_dim = self.d_model
        t.ld_kp, t.ld_qp, t.ld_v = self.key.ld_out[-1], (self.wqk.ld_out[1] if (self.wqk is not None and _QK_CHAIN) else self.key.d_out + 4), self.val.ld_out[-1]
        t.score_act = hip.ACT[self.score_act]
        t.normalize = int(self.normalize)
        t.bkg_score = self.bkg_score
        return t

    def select(self, points, rays_o, rays_d, rays_per_image):
        """Neighbour indices (R,k_eff) int32: kNN, or every point when select_k >= P or < 0
        (reference models/model.py:326-329)."""
        P = points.shape[0]
        k = self.k_cfg
        if k >= P or k < 0:
            R = rays_d.shape[0]
            return torch.arange(P, device=points.device, dtype=torch.int32).expand(R, P).contiguous()
        if k > 255:
            raise NotImplementedError("papr_amd: select_k=%d > 255 is not supported by the HIP kernels" % k)
        return ray_knn(points, rays_o, rays_d, rays_per_image, k, self.eps)


class _RenderFn(torch.autograd.Function):
    """fused (R,C), attn (R,k+1) = path(points, pc_feats, influ, effective weights; rays, idx)."""

    @staticmethod
    def forward(ctx, plan, rays_o, rays_d, rays_per_image, track, idx, points, pc_feats, influ, n_k, n_q, n_v, *wb):
        lib = hip.lib()
        R, k = idx.shape
        M = R * k
        if k > 255:
            raise NotImplementedError("papr_amd: %d neighbours per ray > 255 is not supported by the HIP kernels" % k)
        dev = points.device
        # track: grad mode of the caller (always off in here, and needs_input_grad ignores torch.no_grad()): under no_grad
        # nothing is kept for a backward pass -- no saved activations, no pair sort
        keep = track and (ctx.needs_input_grad[6] or ctx.needs_input_grad[7] or ctx.needs_input_grad[8] or any(ctx.needs_input_grad[12:]))
        # unpack effective weights: key MLP, w_k, query MLP, w_q, value MLP  (each: weights then biases)
        it = iter(wb)
        take = lambda n: [next(it) for _ in range(n)]
        kw, kb = take(n_k), take(n_k)
        wkw, wkb = take(1), take(1)
        qw, qb = take(n_q), take(n_q)
        wqw, wqb = take(1), take(1)
        vw, vb = take(n_v), take(n_v)

        fd = plan.feature_desc(k)
        key_in = torch.empty((M, plan.key.ld_in), device=dev, dtype=torch.float32)
        qry_in = torch.empty((R, plan.qry.ld_in), device=dev, dtype=torch.float32)
        val_in = torch.empty((M, plan.val.ld_in), device=dev, dtype=torch.float32)
        sel = torch.empty((M, 3), device=dev, dtype=torch.float32)
        feats = pc_feats if plan.use_feats else None
        eps = plan.eps
        # the statistics of the LayerNorm core in front of the key MLP come from the kernel that writes the key rows (it has every row in one thread's
        # hands); the fused run applies them while it stages the rows.  PAPR_KEY_STATS=0: the run takes them itself (two wave sums, a square root and a
        # division per row in its staging slot)
        key_given = None
        if plan.k_norm and not plan.fdesc.key_has_feats and _KEY_STATS:
            key_given = (torch.empty((M, 2), device=dev, dtype=torch.float32), torch.empty((M,), device=dev, dtype=torch.float32))
        hip.check(lib.papr_build_features_fwd(C.byref(fd), hip.ptr(points), hip.ptr(feats), hip.ptr(rays_o), hip.ptr(rays_d), R,
                                              rays_per_image, hip.ptr(idx), hip.ptr(key_in), hip.ptr(qry_in), hip.ptr(val_in),
                                              hip.ptr(sel), hip.ptr(key_given[0]) if key_given else None, hip.ptr(key_given[1]) if key_given else None,
                                              eps, hip.stream_ptr()), "papr_build_features_fwd")
        # (the LayerNorm cores in front of and behind the key / query MLPs ride in the fused runs: rows are standardised
        # while they are staged, and again in the last row phase)
        q_outs = mlp_forward(plan.qry, qw, qb, qry_in, R, keep, (plan.qry.d_out, eps) if plan.q_norm else None,
                             (plan.qry_w, eps) if plan.q_norm else None)
        Q = q_outs[-1]
        # score_j = (W_q Q + b_q).(W_k K_j + b_k) = K_j.(W_k^T q') + b_k.q'  with q' = W_q Q + b_q: the R*k-row w_k
        # product of the reference (models/attn.py:217) becomes two R-row products (plain library GEMMs)
        qk_outs = None
        if plan.wqk is not None and _QK_CHAIN:
            # one two-layer run: q' (kept: the backward pass and c0 need it), then g = W_k^T q'; c0 = q'.b_k by one papr_row_dots launch
            wkT = wkw[0].t().contiguous()                     # (kept for the backward pass: one transpose launch per step, not two)
            qk_outs = mlp_forward(plan.wqk, [wqw[0], wkT], [wqb[0], _zeros(dev, (plan.wqk.ld_out[1],))], Q, R, True)
            qk_outs.wkT = wkT
            qp, g = qk_outs[0], qk_outs[1]                   # (R, d_model), (R, key.d_out padded to 32)
            c0 = torch.empty((R,), device=dev, dtype=torch.float32)
            bk = wkb[0].contiguous()
            hip.check(lib.papr_row_dots(hip.ptr(qp), R, plan.d_model, qp.shape[1], hip.ptr(bk), bk.shape[0], max(R, 1), hip.ptr(c0), hip.stream_ptr()), "papr_row_dots")
        else:
            qp = mlp_forward(plan.wq, wqw, wqb, Q, R, True)[0]
            w_aug = torch.cat([wkw[0].t(), wkb[0][None, :], _zeros(dev, (3, wkb[0].shape[0]))], 0).contiguous()
            g = linear_rows(qp, w_aug)                           # (R, key.d_out + 4): [W_k^T q' | b_k.q' | 0 0 0]
            c0 = g[:, plan.key.d_out].contiguous()               # (R,)
        # the dot products K_j.g are taken in the key run's last row phase: in inference the (R*k, d_model) key embedding is never written
        # (1 KB per pair out and back in again otherwise), in training the attention tail does not read it back (the backward pass does);
        # PAPR_SCORES_IN_RUN=0 for the A/B
        in_run = plan.k_norm and _SCORES_IN_RUN
        # training: the key embedding stays RAW in memory (its only reader, the tail's backward pass, standardises what it loads): the fused run's
        # last row phase then takes no row statistics.  Only where the whole key MLP is one fused run (every width a multiple of 32, the fused modes)
        raw_keys = (in_run and keep and _RAW_KEYS and mlp_mode(plan.key.one_product) in (hip.MLP_MODES["h3"], hip.MLP_MODES["h1"], hip.MLP_MODES["h1_f32rows"], hip.MLP_MODES["h3_f16rows"])
                    and 2 <= plan.key.n_layer <= 8 and all(L["n_out"] % 32 == 0 and L["n_out"] <= 256 and not L["skip"] for L in plan.key.layers)
                    and plan.key.last_act == hip.ACT["none"])
        k_outs = mlp_forward(plan.key, kw, kb, key_in, M, keep, (plan.key.d_out, eps) if plan.k_norm else None,
                             ((plan.key_w, eps) + (key_given or ())) if plan.k_norm else None, dot_rows=g if in_run else None, rows_per_dot=k,
                             raw_rows=raw_keys, leave_input=bool(plan.k_norm and key_given is not None and _LN_IN_FEATURES))
        K = k_outs[-1]
        kst, qst, kst2, qst2 = k_outs.in_stats, q_outs.in_stats, k_outs.norm_stats, q_outs.norm_stats
        v_outs = mlp_forward(plan.val, vw, vb, val_in, M, keep, (plan.val.d_out, eps) if plan.v_norm else None,
                             (plan.val_w, eps) if plan.v_norm else None)
        V = v_outs[-1]
        v_aff = None
        if plan.v_norm:                              # the value out-norm's affine a_2 * V^ + b_2 (models/attn.py:42): no Linear behind it to fold into
            va, vbias = wb[-2], wb[-1]
            v_hat = V
            V = torch.zeros_like(v_hat)
            V[:, :plan.val.d_out] = v_hat[:, :plan.val.d_out] * va + vbias
            v_aff = (v_hat, va, V)
        td = plan.tail_desc(k)
        td.precomputed_dots = int(in_run)
        scores = torch.empty((R, k), device=dev, dtype=torch.float32)
        attn = torch.empty((R, k + 1), device=dev, dtype=torch.float32)
        fused = torch.empty((R, plan.C), device=dev, dtype=torch.float32)
        hip.check(lib.papr_attn_tail_fwd(C.byref(td), hip.ptr(k_outs.dots if in_run else K), hip.ptr(g), hip.ptr(c0), hip.ptr(V), hip.ptr(influ), hip.ptr(idx), R,
                                         hip.ptr(scores), hip.ptr(attn), hip.ptr(fused), hip.stream_ptr()), "papr_attn_tail_fwd")
        ctx.plan, ctx.rpi, ctx.n = plan, rays_per_image, (n_k, n_q, n_v)
        ctx.mark_non_differentiable(sel)
        if keep:
            # pairs grouped by selected point, for the atomic-free scatter of the per-point gradients
            order, sorted_pts, seg = group_pairs(idx.view(-1), points.shape[0], run=k)
            ctx.saved = dict(rays_o=rays_o, rays_d=rays_d, idx=idx, points=points, influ=influ, key_in=key_in, qry_in=qry_in,
                             val_in=val_in, kst=kst, qst=qst, kst2=kst2, qst2=qst2, k_outs=k_outs, q_outs=q_outs,
                             v_outs=v_outs, v_aff=v_aff, g=g, c0=c0, qp=qp, qk_outs=qk_outs, key_given=key_given, order=order, sorted_pts=sorted_pts, seg=seg, scores=scores, attn=attn, wb=wb, P=points.shape[0],
                             feat_shape=None if pc_feats is None else pc_feats.shape)
        return fused, attn, sel

    @staticmethod
    def backward(ctx, d_fused, d_attn, _d_sel):
        lib = hip.lib()
        plan, s = ctx.plan, ctx.saved
        n_k, n_q, n_v = ctx.n
        idx = s["idx"]
        R, k = idx.shape
        M = R * k
        dev = idx.device
        eps = plan.eps
        it = iter(s["wb"])
        take = lambda n: [next(it) for _ in range(n)]
        kw, kb = take(n_k), take(n_k)
        wkw, wkb = take(1), take(1)
        qw, qb = take(n_q), take(n_q)
        wqw, wqb = take(1), take(1)
        vw, vb = take(n_v), take(n_v)
        K, Q, V = s["k_outs"][-1], s["q_outs"][-1], s["v_outs"][-1]
        if s["v_aff"] is not None:
            V = s["v_aff"][2]                        # (what the tail multiplied: the value out-norm's affine applied)

        td = plan.tail_desc(k)
        # one-product mode: the key / value gradient rows leave the tail kernel as the data-gradient runs stage them (F16Rows) where kernel and run agree
        need_pts = ctx.needs_input_grad[6]
        need_val_dx = need_pts or ctx.needs_input_grad[7]
        k16 = v16 = None
        if k <= 63 and K.shape[1] == plan.key.d_out == plan.d_model == 256:
            kind = mlp_backward_takes_f16(plan.key, kw, kb, need_pts)
            k16 = F16Rows(dev, M, 256, kind == 2) if kind else None
        if k <= 63 and not plan.v_norm and V.shape[1] == plan.val.d_out and td.C == V.shape[1] and td.C in (4, 8, 16, 32, 64, 128):
            kind = mlp_backward_takes_f16(plan.val, vw, vb, need_val_dx)
            v16 = F16Rows(dev, M, td.C, kind == 2) if kind else None
        global TAIL_F16_ROWS_USED
        TAIL_F16_ROWS_USED += (k16 is not None) + (v16 is not None)
        d_K = None if k16 else (torch.empty_like(K) if K.shape[1] == plan.key.d_out else torch.zeros_like(K))
        d_g = torch.empty_like(s["g"])
        d_c0 = torch.empty((R,), device=dev, dtype=torch.float32)
        d_V = None if v16 else torch.empty_like(V)
        # the three per-point gradients share ONE zeroed buffer (points without a pair keep the zeros; one fill launch instead of three):
        # [features | points | influence], the features first so that their rows stay 16-byte aligned
        n_pts = s["P"]
        n_feat = 0
        if s["feat_shape"] is not None:
            n_feat = 1
            for v in s["feat_shape"]:
                n_feat *= int(v)
        per_point = torch.zeros((n_feat + 4 * n_pts,), device=dev, dtype=torch.float32)
        d_influ = per_point[n_feat + 3 * n_pts:].view(n_pts, 1)
        pair_influ = torch.empty((M,), device=dev, dtype=torch.float32)
        d_fused = d_fused.contiguous()
        d_attn = d_attn.contiguous() if d_attn is not None else None
        # (with the LayerNorm core behind the key MLP, d_K comes back as the gradient in front of it: no papr_rownorm_bwd pass)
        hip.check(lib.papr_attn_tail_bwd(C.byref(td), hip.ptr(K), hip.ptr(s["g"]), hip.ptr(V), hip.ptr(s["influ"]), hip.ptr(idx),
                                         R, hip.ptr(s["scores"]), hip.ptr(s["attn"]), hip.ptr(d_fused), hip.ptr(d_attn), hip.ptr(d_K),
                                         hip.ptr(d_g), hip.ptr(d_V), None, hip.ptr(d_c0), hip.ptr(pair_influ),
                                         hip.ptr(s["kst2"]) if plan.k_norm else None, hip.ptr(s["c0"]) if plan.k_norm else None,
                                         hip.ptr(s["k_outs"].norm_mean), k16.ref() if k16 else None, v16.ref() if v16 else None, hip.stream_ptr()), "papr_attn_tail_bwd")
        # backward of g = q' W_k, c0 = q'.b_k: R-row products on the library's GEMMs (rocBLAS / hipBLASLt pick 130-270 us
        # kernels for these 25,600 x 256 shapes; the same work is ~100 us here)
        qp = s["qp"]
        wmax = max(plan.key.width, plan.qry.width, plan.val.width, plan.key.ld_in, plan.val.ld_in, plan.d_model)
        scratch = [torch.empty((M, wmax), device=dev, dtype=torch.float32) for _ in range(2)]
        fused_qk_bwd = s["qk_outs"] is not None and _QK_CHAIN >= 2
        if fused_qk_bwd:
            # the two layers' data- and weight-gradients in one run each; the score bias c0 = q'.b_k = Q.(W_q^T b_k) + b_q.b_k hangs on q' BETWEEN the
            # two layers of that run, so its share reaches W_q, b_q and Q as rank-one terms (outer products of R-row sums, a handful of small launches)
            if d_g.shape[1] != plan.wqk.ld_out[1] or not d_g.is_contiguous():
                d_g = d_g[:, :plan.wqk.ld_out[1]].contiguous()
            qs = [t[:R] for t in scratch]
            (d_wq_a, d_wkT), (d_wqb_a, _), d_Q = mlp_backward(plan.wqk, [wqw[0], s["qk_outs"].wkT], [wqb[0], _zeros(dev, (plan.wqk.ld_out[1],))],
                                                              Q, R, s["qk_outs"], d_g, qs, True)
            bk = wkb[0].contiguous()
            if _QK_BIAS_KERNEL:                      # papr_qk_bias_bwd: the five rank-one terms in two launches
                d_wkb_v = torch.empty((plan.d_model,), device=dev, dtype=torch.float32)
                ws = torch.empty((lib.papr_qk_bias_bwd_workspace_bytes(plan.qry.d_out) + 3) // 4, device=dev, dtype=torch.float32)
                bq_ = wqb[0].contiguous()
                # (raw pointers go in: the preconditions are checked with a raise, not an assert -- `python -O` must not turn a padded or transposed
                #  W_q into a read with the wrong pitch; the weight's leading dimension is its row stride.  ADVICE r05)
                wq0 = wqw[0] if wqw[0].stride(1) == 1 else wqw[0].contiguous()
                if not (d_wqb_a.shape[0] == plan.d_model and bk.shape[0] >= plan.d_model and d_Q.is_contiguous() and d_wq_a.is_contiguous()
                        and d_wq_a.shape == wq0.shape and Q.stride(1) == 1 and d_Q.shape == Q.shape):
                    raise RuntimeError("papr_amd: papr_qk_bias_bwd: d_Q / d_W_q must be contiguous and shaped like Q / W_q, b_k at least d_model long")
                hip.check(lib.papr_qk_bias_bwd(hip.ptr(Q), Q.stride(0), plan.qry.d_out, plan.d_model, hip.ptr(d_c0), R, hip.ptr(wq0), wq0.stride(0),
                                               hip.ptr(bk), hip.ptr(bq_), hip.ptr(d_Q), hip.ptr(d_wq_a), hip.ptr(d_wqb_a), hip.ptr(d_wqb_a), hip.ptr(d_wkb_v),
                                               hip.ptr(ws), hip.stream_ptr()), "papr_qk_bias_bwd")
                d_wq, d_wqb = [d_wq_a], [d_wqb_a]
            else:                                    # (A/B: the same in torch ops, ten launches)
                both = torch.cat([Q[:, :plan.qry.d_out], qp[:, :plan.d_model]], 1)                  # one reduction for Q^T d_c0 and q'^T d_c0
                red = (both * d_c0[:, None]).sum(0)
                u, d_wkb_v = red[:plan.qry.d_out], red[plan.qry.d_out:]
                vq = (wqw[0][:, :plan.qry.d_out] * bk[:plan.d_model, None]).sum(0)                      # W_q^T b_k
                d_Q[:, :plan.qry.d_out].addcmul_(d_c0[:, None], vq[None, :])
                d_wq_a[:, :plan.qry.d_out].addcmul_(bk[:plan.d_model, None], u[None, :])
                d_wq, d_wqb = [d_wq_a], [torch.addcmul(d_wqb_a, bk[:d_wqb_a.shape[0]], d_c0.sum())]
            d_wk, d_wkb = [d_wkT.t()], [d_wkb_v.contiguous()]
        else:
            d_g = d_g[:, :plan.key.d_out].contiguous()           # the tail kernel fills d_model = key.d_out columns
            d_wkT, _, d_qp = mlp_backward(plan.wk_fold, [wkw[0].t().contiguous()], [_zeros(dev, (plan.key.d_out,))], qp, R, [d_g], d_g,
                                          [t[:R] for t in scratch], True)
            d_qp.addcmul_(d_c0[:, None], wkb[0][None, :])
            d_wk = [d_wkT[0].t()]
            d_wkb = [(qp * d_c0[:, None]).sum(0)]               # (two launches, 38 us; torch.mv(qp.t(), d_c0) lands on a 269-us rocBLAS gemv kernel: measured, reverted)
        # key branch
        d_kw, d_kb, d_key = mlp_backward(plan.key, kw, kb, s["key_in"], M, s["k_outs"], k16 or d_K, scratch, need_pts)
        # the backward pass of the LayerNorm core in front of the key MLP rides in papr_build_features_bwd_pairs where that kernel has the row's
        # statistics from the forward pass (key_given: the default); otherwise one papr_rownorm_bwd pass over the gradient rows
        ln_in_features = plan.k_norm and d_key is not None and s["key_given"] is not None and _LN_IN_FEATURES
        if plan.k_norm and d_key is not None and not ln_in_features:
            rownorm_bwd_(d_key, s["key_in"], s["kst"], plan.key_w, eps)
        # query branch (the ray directions need no gradient)
        qscratch = [t[:R] for t in scratch]
        if not fused_qk_bwd:
            d_wq, d_wqb, d_Q = mlp_backward(plan.wq, wqw, wqb, Q, R, [qp], d_qp.contiguous(), qscratch, True)
        if plan.q_norm:
            rownorm_bwd_(d_Q, Q, s["qst2"], plan.qry.d_out, eps)
        d_qw, d_qb, _ = mlp_backward(plan.qry, qw, qb, s["qry_in"], R, s["q_outs"], d_Q, qscratch, False)
        # value branch
        d_va = d_vbias = None
        if plan.v_norm:                              # back through the value out-norm: its affine (host glue), then its core
            v_hat, va, _ = s["v_aff"]
            dv = d_V[:, :plan.val.d_out]
            d_va, d_vbias = (dv * v_hat[:, :plan.val.d_out]).sum(0), dv.sum(0)
            d_hat = torch.zeros_like(d_V)
            d_hat[:, :plan.val.d_out] = dv * va
            d_V = rownorm_bwd_(d_hat, v_hat, s["v_outs"].norm_stats, plan.val.d_out, eps)
        d_vw, d_vb, d_val = mlp_backward(plan.val, vw, vb, s["val_in"], M, s["v_outs"], v16 or d_V, scratch, need_val_dx)
        if plan.v_norm and d_val is not None:        # ... and its in-norm's core (the affine is folded into the first layer's weights)
            rownorm_bwd_(d_val, s["val_in"], s["v_outs"].in_stats, plan.val_w, eps)
        # gather / geometry / encoding backward: per-pair gradient rows, then one segmented sum per point
        d_points = d_feats = pair_pts = None
        need_geo = need_pts or ctx.needs_input_grad[7]
        if need_geo:
            pair_pts = torch.empty((M, 4), device=dev, dtype=torch.float32)
            fd = plan.feature_desc(k)
            hip.check(lib.papr_build_features_bwd_pairs(C.byref(fd), hip.ptr(s["points"]), hip.ptr(s["rays_o"]), hip.ptr(s["rays_d"]), R,
                                                        ctx.rpi, hip.ptr(idx), hip.ptr(d_key), hip.ptr(d_val), hip.ptr(pair_pts),
                                                        hip.ptr(s["key_given"][1]) if ln_in_features else None,
                                                        hip.ptr(s["key_given"][0]) if ln_in_features else None,
                                                        hip.stream_ptr()), "papr_build_features_bwd_pairs")
            d_points = per_point[n_feat:n_feat + 3 * n_pts].view(n_pts, 3)
        fdim = plan.feat_dim
        rows, ld, col0 = None, 0, 0
        if need_geo and plan.use_feats:
            d_feats = per_point[:n_feat].view(s["feat_shape"])
            if plan.fdesc.val_has_feats:
                rows, ld, col0 = d_val, d_val.shape[1], plan.val_w - fdim
            else:
                rows, ld, col0 = d_key, d_key.shape[1], plan.key_w - fdim
        seg_ws = torch.empty(lib.papr_segment_reduce_workspace_bytes(M) // 4, device=dev, dtype=torch.float32)
        hip.check(lib.papr_segment_reduce(hip.ptr(s["order"]), hip.ptr(s["sorted_pts"]), hip.ptr(s["seg"]), M, s["P"], hip.ptr(pair_pts), hip.ptr(pair_influ),
                                          hip.ptr(rows), ld, col0, fdim if rows is not None else 0, hip.ptr(d_points), hip.ptr(d_influ),
                                          hip.ptr(d_feats), 0, hip.ptr(seg_ws), hip.stream_ptr()), "papr_segment_reduce")
        if need_geo and plan.fdesc.val_has_feats and plan.fdesc.key_has_feats:      # features feed both branches
            hip.check(lib.papr_segment_reduce(hip.ptr(s["order"]), hip.ptr(s["sorted_pts"]), hip.ptr(s["seg"]), M, s["P"], None, None,
                                              hip.ptr(d_key), d_key.shape[1], plan.key_w - fdim, fdim, None, None, hip.ptr(d_feats),
                                              1, hip.ptr(seg_ws), hip.stream_ptr()), "papr_segment_reduce")
        ctx.saved = None
        grads_wb = d_kw + d_kb + d_wk + d_wkb + d_qw + d_qb + d_wq + d_wqb + d_vw + d_vb + ([d_va, d_vbias] if plan.v_norm else [])
        return (None, None, None, None, None, None, d_points, d_feats, d_influ, None, None, None) + tuple(grads_wb)


def render_rays(plan, rays_o, rays_d, rays_per_image, idx, points, pc_feats, influ, weights):
    """weights: dict with key/wk/query/wq/value -> (list of W, list of b) already prepared for the kernels."""
    kw, kb = weights["key"]
    qw, qb = weights["query"]
    vw, vb = weights["value"]
    flat = kw + kb + weights["wk"][0] + weights["wk"][1] + qw + qb + weights["wq"][0] + weights["wq"][1] + vw + vb
    if plan.v_norm:
        flat = flat + list(weights["v_out"])         # (a_2, b_2) of the value out-norm
    return _RenderFn.apply(plan, rays_o, rays_d, rays_per_image, torch.is_grad_enabled(), idx, points, pc_feats, influ, len(kw), len(qw), len(vw), *flat)
