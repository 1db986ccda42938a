"""The output activation `models.last_act` (reference models/model.py:104 -> models/utils.py:183-229 activation_func with its default
arguments a = b = 1, trainable = False, num_channels = 128, neg_slope = 0.2): an elementwise function of the composited RGB, applied by the
drivers (train.py:170, test.py:103) -- outside the kernels, so plain torch.  Every name the reference accepts is accepted; the ones with a
shape constant (`gaussian`, `quadratic`, ...) hold it as a frozen parameter `a` (`b`) so that the state-dict keys match (`last_act.a`)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _Fn(nn.Module):
    def __init__(self, fn, consts=()):
        super().__init__()
        self.fn = fn
        for name in consts:                       # (frozen, like the reference's `trainable=False` default)
            self.register_parameter(name, nn.Parameter(torch.ones(1), requires_grad=False))

    def forward(self, x):
        return self.fn(self, x)


_TABLE = {
    "+1": (lambda m, x: x + 1, ()),
    "relu+1": (lambda m, x: torch.relu(x) + 1, ()),
    "tanh": (lambda m, x: torch.tanh(x), ()),
    "shifted_tanh": (lambda m, x: (torch.tanh(x) + 1) / 2, ()),
    "sigmoid": (lambda m, x: torch.sigmoid(x), ()),
    "gelu": (lambda m, x: F.gelu(x), ()),
    "clamp": (lambda m, x: torch.clamp(x, 0, 1), ()),
    "gaussian": (lambda m, x: torch.exp(-x ** 2 / (2 * m.a ** 2)), ("a",)),
    "quadratic": (lambda m, x: 1 / (1 + (m.a * x) ** 2), ("a",)),
    "multi-quadratic": (lambda m, x: 1 / (1 + (m.a * x) ** 2) ** 0.5, ("a",)),
    "laplacian": (lambda m, x: torch.exp(-torch.abs(x) / m.a), ("a",)),
    "super-gaussian": (lambda m, x: torch.exp(-x ** 2 / (2 * m.a ** 2)) ** m.b, ("a", "b")),
    "expsin": (lambda m, x: torch.exp(-torch.sin(m.a * x)), ("a",)),
}


def output_activation(name):
    key = name.lower()
    if key == "none":
        return nn.Identity()
    if key == "relu":
        return nn.ReLU()
    if key == "leakyrelu":
        return nn.LeakyReLU(0.2)
    if key == "prelu":
        return nn.PReLU(128)
    if key in _TABLE:
        fn, consts = _TABLE[key]
        return _Fn(fn, consts)
    if "sine" in key:                             # sin(a x), a = 1
        return _Fn(lambda m, x: torch.sin(x * 1.0))
    if "softplus" in key:                         # "softplus_<c1>_<c2>_<c3>": c1 softplus(c2 x + c3)
        c1, c2, c3 = (float(v) for v in key.split("_")[1:])
        print("Softplus activation: a={:.2f}, b={:.2f}, c={:.2f}".format(c1, c2, c3))
        return _Fn(lambda m, x: c1 * F.softplus(c2 * x + c3))
    raise NotImplementedError("activation layer [{:s}] is not found".format(key))
