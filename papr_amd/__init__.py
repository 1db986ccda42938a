"""papr_amd -- MI355X-native PAPR render path (HIP kernels behind the PAPR nn.Module surface)."""
from .config import load_config, ConfigNode, deep_merge, eval_config  # noqa: F401


def get_model(args, device="cuda"):
    """Counterpart of the reference's models.get_model (models/__init__.py:23-24)."""
    from .model import PAPR
    return PAPR(args, device=device)


def get_loss(args, bias=1.0):
    """Counterpart of the reference's models.get_loss (models/__init__.py:27-52)."""
    from .loss import get_loss as _get
    return _get(args, bias)
