"""Training losses (counterpart of the reference's BasicLoss / get_loss, models/__init__.py:8-52).

MSE, L1 and the VGG16 LPIPS term (papr_amd/lpips.py: the reference's LPNet) are supported.  LPIPS needs pretrained
weights that the reference downloads through torchvision; they cannot be fetched here, so the loss loads them from
user-supplied files and raises -- rather than silently training another objective -- when they are absent
(`--set training.losses.lpips=0` trains MSE-only; PSNRs of such runs are labelled MSE-only).  `lpips_alex` (the `lpips`
package's AlexNet network, weight 0 in every shipped config) is not built.
"""
import os

import torch
import torch.nn as nn

_OWN_MSE = os.environ.get("PAPR_OWN_MSE", "1") == "1"          # (0: torch.nn.MSELoss on the device too; A/B)


_mse_ws = {}


class _MseFn(torch.autograd.Function):
    """mean((pred - target)^2) and its gradient direction 2 (pred - target) / n in one launch (papr_mse_fwd); backward: one multiply."""

    @staticmethod
    def forward(ctx, pred, target):
        from . import hip
        pred, target = pred.contiguous(), target.contiguous()
        loss = torch.empty((), device=pred.device, dtype=torch.float32)
        grad = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        key = (pred.device.index, torch.cuda.current_stream(pred.device).cuda_stream)        # (one ticket workspace per device and stream -- of pred's device)
        ws = _mse_ws.get(key)
        if ws is None:
            ws = _mse_ws[key] = torch.zeros(hip.lib().papr_mse_workspace_bytes(), device=pred.device, dtype=torch.uint8)
        hip.check(hip.lib().papr_mse_fwd(hip.ptr(pred), hip.ptr(target), pred.numel(), hip.ptr(loss), hip.ptr(grad), hip.ptr(ws), hip.stream_ptr()), "papr_mse_fwd")
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, d_loss):
        grad, = ctx.saved_tensors
        return grad * d_loss, None


class MSELoss(nn.MSELoss):
    """torch.nn.MSELoss() (the reference's `mse` term, models/__init__.py:21-22); fp32 device tensors of a training patch's size take the library's
    one-launch kernel (the target needs no gradient there), everything else torch's."""

    def forward(self, pred, target):
        if (_OWN_MSE and self.reduction == "mean" and pred.is_cuda and pred.dtype == torch.float32 and target.dtype == torch.float32 and pred.shape == target.shape
                and not target.requires_grad and 0 < pred.numel() <= (1 << 22)):
            return _MseFn.apply(pred, target)
        return super().forward(pred, target)


class BasicLoss(nn.Module):
    def __init__(self, terms):
        super().__init__()
        self.terms = terms

    def forward(self, pred, target):
        # total = 0 + w_1 * term_1 + ... as the reference writes it; the launches that change nothing (0 + x, 1.0 * x) are not issued
        total = None
        for tag, fn in self.terms.items():
            w = float(tag.split("/")[1])
            term = fn(pred, target)
            term = term if w == 1.0 else w * term
            total = term if total is None else total + term
        return 0 if total is None else total


def get_loss(args, bias=1.0):
    terms = nn.ModuleDict()
    for name, weight in dict(args).items():
        if weight <= 0:
            continue
        tag = name + "/" + format(weight, ".0e")
        if name == "mse":
            terms[tag] = MSELoss()
        elif name == "l1":
            terms[tag] = nn.L1Loss()
        elif name == "lpips":
            from .lpips import LPNet
            net = LPNet()                 # raises FileNotFoundError with instructions when the weight files are absent
            net.eval()
            terms[tag] = net
        elif name == "lpips_alex":
            raise NotImplementedError("papr_amd: loss 'lpips_alex' needs the `lpips` package's AlexNet weights, which are not "
                                      "available offline; set training.losses.lpips_alex to 0")
        else:
            raise NotImplementedError("loss [%s] is not supported" % name)
    return BasicLoss(terms)
