"""Training losses (counterpart of the reference's BasicLoss / get_loss, models/__init__.py:8-52).

MSE and L1 are supported.  The LPIPS terms need VGG16 / AlexNet ImageNet weights that the reference
downloads through torchvision / lpips; neither package nor the weights exist offline, so a non-zero
LPIPS weight is an explicit error rather than a silently different loss.
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
        elif name in ("lpips", "lpips_alex"):
            raise NotImplementedError("papr_amd: loss '%s' needs pretrained VGG/AlexNet weights that are not available "
                                      "offline; set training.losses.%s to 0" % (name, name))
        else:
            raise NotImplementedError("loss [%s] is not supported" % name)
    return BasicLoss(terms)
