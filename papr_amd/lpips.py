"""LPIPS (VGG16) training loss -- counterpart of the reference's LPNet (models/lpips.py:8-125).

The loss is not on the per-ray render path and runs on torch convolutions; it exists so that the
scene files' own `training.losses.lpips: 0.01` trains here as it does in the reference.  The arithmetic follows
models/lpips.py:95-125: images (N,H,W,3) in [0,1] -> 2x-1 -> channel shift / scale -> VGG16 features after relu1_2, 2_2,
3_3, 4_3, 5_3 -> unit-normalised over channels -> squared difference -> learnt per-channel weights -> spatial mean -> sum
over the five taps -> mean over the batch.

Weights -- neither ships with this package (no network here, and the backbone is 56 MB):
  * the five linear heads: `vgg.pth` in the working directory, exactly where the reference looks for it
    (models/lpips.py:95-101; the file ships in the reference's repository root), or $PAPR_LPIPS_HEADS;
  * the VGG16 ImageNet backbone the reference pulls through torchvision (`VGG16_Weights.IMAGENET1K_V1`): a torchvision
    state dict (`features.N.weight/bias`, e.g. vgg16-397923af.pth) named by $PAPR_VGG16_WEIGHTS or `vgg16.pth` in the working
    directory.
If either is missing, building the loss raises with these instructions instead of training a different objective.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

# torchvision's vgg16().features: index -> ("conv", c_in, c_out) | "relu" | "pool"; taps after indices 3, 8, 15, 22, 29
_VGG16 = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512]
_TAPS = (3, 8, 15, 22, 29)


def vgg16_features():
    """nn.Sequential with torchvision's layer indices (so that its state-dict keys load unchanged), up to relu5_3."""
    layers, c_in = [], 3
    for v in _VGG16:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(c_in, v, kernel_size=3, padding=1), nn.ReLU(inplace=False)]
            c_in = v
    return nn.Sequential(*layers)


class LPNet(nn.Module):
    def __init__(self, backbone_path=None, heads_path=None, load=True):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])
        self.features = vgg16_features()
        self.lins = nn.ParameterList([nn.Parameter(torch.zeros(1, c, 1, 1)) for c in (64, 128, 256, 512, 512)])
        if load:
            self._load(backbone_path, heads_path)
        for p in self.features.parameters():                 # (models/lpips.py:28-30: the backbone is frozen, the heads are not)
            p.requires_grad = False

    def _load(self, backbone_path, heads_path):
        heads_path = heads_path or os.environ.get("PAPR_LPIPS_HEADS") or os.path.abspath(os.path.join(".", "vgg.pth"))
        backbone_path = backbone_path or os.environ.get("PAPR_VGG16_WEIGHTS") or os.path.abspath(os.path.join(".", "vgg16.pth"))
        missing = [p for p in (heads_path, backbone_path) if not os.path.exists(p)]
        if missing:
            raise FileNotFoundError(
                "papr_amd: the LPIPS loss needs pretrained weights that cannot be fetched here: %s not found.  Put the reference's "
                "vgg.pth (LPIPS linear heads) in the working directory or name it in $PAPR_LPIPS_HEADS, and name a torchvision VGG16 "
                "state dict (vgg16-397923af.pth) in $PAPR_VGG16_WEIGHTS -- or train MSE-only with "
                "`--set training.losses.lpips=0`" % ", ".join(missing))
        print("Loading model from: %s" % heads_path)
        heads = torch.load(heads_path, map_location="cpu")
        for i, p in enumerate(self.lins):
            p.data.copy_(heads["lin%d.model.1.weight" % i])
        sd = torch.load(backbone_path, map_location="cpu")
        sd = {k[len("features."):]: v for k, v in sd.items() if k.startswith("features.")}
        self.features.load_state_dict(sd)

    def _taps(self, x):
        outs = []
        for i, layer in enumerate(self.features):
            x = layer(x)
            if i in _TAPS:
                outs.append(x)
        return outs

    def forward(self, in0, in1):
        in0 = (2 * in0.permute(0, 3, 1, 2) - 1 - self.shift) / self.scale
        in1 = (2 * in1.permute(0, 3, 1, 2) - 1 - self.shift) / self.scale
        val = None
        for w, f0, f1 in zip(self.lins, self._taps(in0), self._taps(in1)):
            n0 = f0 / (torch.sqrt(torch.sum(f0 ** 2, dim=1, keepdim=True) + 1e-10) + 1e-10)
            n1 = f1 / (torch.sqrt(torch.sum(f1 ** 2, dim=1, keepdim=True) + 1e-10) + 1e-10)
            r = torch.sum(w * (n0 - n1) ** 2, 1, keepdim=True).mean([2, 3], keepdim=True)
            val = r if val is None else val + r
        return val.squeeze().mean()


# ---- evaluation metric of test.py (reference test.py:109-110, 188-191: `lpips.LPIPS(net='alex' | 'vgg', version='0.1')` of the
# `lpips` package, called on rgb / img in [0, 1] WITHOUT normalize=True, i.e. the images go in as they are)
_ALEX = [("conv", 3, 64, 11, 4, 2), "relu", ("pool", 3, 2), ("conv", 64, 192, 5, 1, 2), "relu", ("pool", 3, 2),
         ("conv", 192, 384, 3, 1, 1), "relu", ("conv", 384, 256, 3, 1, 1), "relu", ("conv", 256, 256, 3, 1, 1), "relu"]
_ALEX_TAPS = (1, 4, 7, 9, 11)                   # torchvision alexnet().features indices behind each ReLU


def alexnet_features():
    layers = []
    for v in _ALEX:
        if v == "relu":
            layers.append(nn.ReLU(inplace=False))
        elif v[0] == "pool":
            layers.append(nn.MaxPool2d(kernel_size=v[1], stride=v[2]))
        else:
            layers.append(nn.Conv2d(v[1], v[2], kernel_size=v[3], stride=v[4], padding=v[5]))
    return nn.Sequential(*layers)


class TestLPIPS(nn.Module):
    """LPIPS v0.1 as the `lpips` package evaluates it (Zhang et al. 2018): scaling layer, backbone taps, channel-unit-normalised
    features, squared difference, learnt 1x1 heads, spatial mean, sum over taps.  Weights (none ship here, no network):
      vgg : heads $PAPR_LPIPS_HEADS or ./vgg.pth (the package's weights/v0.1/vgg.pth = the reference's vgg.pth), backbone
            $PAPR_VGG16_WEIGHTS or ./vgg16.pth (torchvision state dict);
      alex: heads $PAPR_LPIPS_HEADS_ALEX or ./alex.pth (weights/v0.1/alex.pth), backbone $PAPR_ALEX_WEIGHTS or ./alexnet.pth
            (torchvision alexnet-owt-7be5be79.pth)."""
    FILES = {"vgg": (("PAPR_LPIPS_HEADS", "vgg.pth"), ("PAPR_VGG16_WEIGHTS", "vgg16.pth")),
             "alex": (("PAPR_LPIPS_HEADS_ALEX", "alex.pth"), ("PAPR_ALEX_WEIGHTS", "alexnet.pth"))}

    def __init__(self, net, heads=None, backbone=None):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])
        self.features = vgg16_features() if net == "vgg" else alexnet_features()
        self.taps = _TAPS if net == "vgg" else _ALEX_TAPS
        chans = (64, 128, 256, 512, 512) if net == "vgg" else (64, 192, 384, 256, 256)
        self.lins = nn.ParameterList([nn.Parameter(torch.zeros(1, c, 1, 1), requires_grad=False) for c in chans])
        if heads is not None:
            for i, p in enumerate(self.lins):
                p.data.copy_(heads["lin%d.model.1.weight" % i])
        if backbone is not None:
            self.features.load_state_dict({k[len("features."):]: v for k, v in backbone.items() if k.startswith("features.")})
        self.eval()

    @classmethod
    def paths(cls, net):
        return [os.environ.get(env) or os.path.abspath(os.path.join(".", default)) for env, default in cls.FILES[net]]

    @classmethod
    def try_build(cls, net, device):
        heads, backbone = cls.paths(net)
        if not (os.path.exists(heads) and os.path.exists(backbone)):
            return None
        return cls(net, torch.load(heads, map_location="cpu"), torch.load(backbone, map_location="cpu")).to(device)

    @torch.no_grad()
    def forward(self, in0, in1):
        """in0, in1: (N, 3, H, W); returns (N,) distances."""
        f0, f1 = (in0 - self.shift) / self.scale, (in1 - self.shift) / self.scale
        val = 0
        for i, layer in enumerate(self.features):
            f0, f1 = layer(f0), layer(f1)
            if i in self.taps:
                w = self.lins[self.taps.index(i)]
                n0 = f0 / (torch.sqrt(torch.sum(f0 ** 2, dim=1, keepdim=True)) + 1e-10)
                n1 = f1 / (torch.sqrt(torch.sum(f1 ** 2, dim=1, keepdim=True)) + 1e-10)
                val = val + torch.sum(w * (n0 - n1) ** 2, 1, keepdim=True).mean([2, 3])
        return val.reshape(-1)
