"""The ONE switch behind which a device tensor of the render head may reach torch's own layers (aten / MIOpen).

The product path has no fallback: on a ROCm device every layer of the render head runs on this library's kernels, and a shape outside their
constraints raises by name -- as the per-ray path does (papr_amd/ops.py, papr_amd/model.py).  For A/B measurements and debugging,

    PAPR_DEBUG_TORCH_HEAD=<part>[,<part>...]     parts: conv | rest | wgrad | autocast | all

hands the named parts of the head to torch, announcing each part once on stderr when it first engages:
  conv      the 3x3 convolution stages (nn.Conv2d + ReLU)
  rest      max-pooling, the 2x2 transposed convolutions and the 1x1 output layer
  wgrad     the 3x3 layers' weight / data gradients through aten.convolution_backward where the own kernels have no form
  autocast  under use_amp: the whole head under torch.autocast (the reference's own arrangement, models/unet.py:212)
(Rounds 3-5 had four import-time switches -- PAPR_UNET_CONV / _REST / _AMP / _WGRAD -- and shape tests that routed to torch without a word.)
A CPU tensor is not the product path: the modules are plain torch there (the float64 reference of the GPU tests)."""
import os
import sys

PARTS = ("conv", "rest", "wgrad", "autocast")
_ON = frozenset(p for p in os.environ.get("PAPR_DEBUG_TORCH_HEAD", "").replace(" ", "").split(",") if p)
_bad = _ON - set(PARTS) - {"all"}
if _bad:
    raise ValueError("PAPR_DEBUG_TORCH_HEAD: unknown part(s) %s (conv | rest | wgrad | autocast | all)" % sorted(_bad))
_said = set()


def torch_head(part):
    """True: PAPR_DEBUG_TORCH_HEAD hands `part` of the render head to torch (announced once)."""
    assert part in PARTS, part
    if "all" in _ON or part in _ON:
        if part not in _said:
            _said.add(part)
            print("papr_amd: PAPR_DEBUG_TORCH_HEAD=%s: the render head's '%s' part runs on torch / MIOpen, NOT on this library's kernels "
                  "(debug / A-B only)" % (os.environ.get("PAPR_DEBUG_TORCH_HEAD"), part), file=sys.stderr, flush=True)
        return True
    return False


def own_or_raise(ok, part, layer, constraint):
    """On a device tensor: True = run `layer` on the own kernels.  Its shape outside their constraints (`ok` false) raises by name unless the debug
    switch hands `part` to torch; with the switch on the part goes to torch whatever the shape."""
    if torch_head(part):
        return False
    if not ok:
        raise NotImplementedError("papr_amd: %s: the own kernels need %s; there is no torch fallback on the device "
                                  "(PAPR_DEBUG_TORCH_HEAD=%s for an A/B run on torch / MIOpen)" % (layer, constraint, part))
    return True
