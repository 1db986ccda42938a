"""Training losses (counterpart of the reference's BasicLoss / get_loss, models/__init__.py:8-52).

MSE, L1 and the VGG16 LPIPS term (papr_amd/lpips.py: the reference's LPNet) are supported.  LPIPS needs pretrained
weights that the reference downloads through torchvision; they cannot be fetched here, so the loss loads them from
user-supplied files and raises -- rather than silently training another objective -- when they are absent
(`--set training.losses.lpips=0` trains MSE-only; PSNRs of such runs are labelled MSE-only).  `lpips_alex` (the `lpips`
package's AlexNet network, weight 0 in every shipped config) is not built.
"""
import torch.nn as nn


class BasicLoss(nn.Module):
    def __init__(self, terms):
        super().__init__()
        self.terms = terms

    def forward(self, pred, target):
        total = 0
        for tag, fn in self.terms.items():
            total = total + float(tag.split("/")[1]) * fn(pred, target)
        return total


def get_loss(args, bias=1.0):
    terms = nn.ModuleDict()
    for name, weight in dict(args).items():
        if weight <= 0:
            continue
        tag = name + "/" + format(weight, ".0e")
        if name == "mse":
            terms[tag] = nn.MSELoss()
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
