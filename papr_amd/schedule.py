"""Learning-rate schedules of the PAPR optimizers.

`create_learning_rate_fn` builds the same torch scheduler objects as the reference
(models/utils.py:260-322: SequentialLR(LinearLR warm-up, decay) with the `verbose` keyword dropped,
which recent torch no longer accepts), so `schedulers.pth` checkpoints stay interchangeable.
`lr_at` is the closed form of that composite, used to fast-forward and by the tests.
"""
import math

from torch.optim import lr_scheduler


def create_learning_rate_fn(optimizer, max_steps, args, debug=False):
    kind = args["type"]
    if kind == "none":
        return None
    warmup = args["warmup"]
    start = 1e-16 if warmup > 0 else 1.0
    warm = lr_scheduler.LinearLR(optimizer, start_factor=start, end_factor=1.0, total_iters=warmup)
    if kind == "linear":
        decay = lr_scheduler.LinearLR(optimizer, start_factor=1.0, end_factor=0.0, total_iters=max_steps - warmup)
    elif kind == "cosine":
        decay = lr_scheduler.CosineAnnealingLR(optimizer, T_max=max(max_steps - warmup, 1))
    elif kind == "cosine-hlfperiod":
        decay = lr_scheduler.CosineAnnealingLR(optimizer, T_max=max(max_steps - warmup, 1) * 2)
    elif kind == "exp":
        decay = lr_scheduler.ExponentialLR(optimizer, gamma=args["gamma"])
    elif kind == "stop":
        decay = lr_scheduler.StepLR(optimizer, step_size=1, gamma=0.0)
    else:
        raise NotImplementedError
    return lr_scheduler.SequentialLR(optimizer, schedulers=[warm, decay], milestones=[warmup])


def lr_at(args, max_steps, step, lr_factor=1.0):
    """Learning rate in effect for optimizer step number `step` (0-based) under the schedule above."""
    base = args["base_lr"] * lr_factor
    kind = args["type"]
    if kind == "none":
        return base
    warmup = args["warmup"]
    if step < warmup:
        start = 1e-16
        return base * (start + (1.0 - start) * step / warmup)
    t = step - warmup
    if kind == "linear":
        total = max_steps - warmup
        return base * (1.0 - min(t, total) / total) if total > 0 else base
    if kind == "cosine":
        return base * (1 + math.cos(math.pi * t / max(max_steps - warmup, 1))) / 2
    if kind == "cosine-hlfperiod":
        return base * (1 + math.cos(math.pi * t / (max(max_steps - warmup, 1) * 2))) / 2
    if kind == "exp":
        return base * args["gamma"] ** t
    if kind == "stop":
        return base if t == 0 else 0.0
    raise NotImplementedError
