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
    if kind in ("cosine", "cosine-hlfperiod"):
        T = max(max_steps - warmup, 1) * (2 if kind == "cosine-hlfperiod" else 1)
        if warmup == 0:
            # torch's SequentialLR rewinds every member by one epoch at construction and only re-runs the first one's
            # initial step, so without a warm-up the cosine member starts at epoch -1: its first step multiplies the
            # base lr by 2 / (1 + cos(pi / T)) and the curve runs one step late.  (Same in the reference's torch.)
            return base if step == 0 else base * (1 + math.cos(math.pi * (step - 1) / T)) / (1 + math.cos(math.pi / T))
        return base * (1 + math.cos(math.pi * t / T)) / 2
    if kind == "exp":
        return base * args["gamma"] ** t
    if kind == "stop":
        return base if t == 0 else 0.0
    raise NotImplementedError


def fast_forward(sched, args, max_steps, n, lr_factor=1.0):
    """Put a freshly created schedule (create_learning_rate_fn) into the state n `sched.step()` calls would leave it in.

    The reference re-creates optimizers and schedules at every prune / add event and then replays the schedule step by
    step (models/model.py:175-179): an O(step) Python loop, five schedules, every 500 steps.  torch's LinearLR and
    CosineAnnealingLR advance recursively from (current lr, last_epoch) alone, so setting those two to their closed-form
    values (lr_at) is equivalent; tests/test_host_model.py compares against the loop."""
    if sched is None or n <= 0:
        return
    warmup = args["warmup"]
    if warmup == 0 and args["type"] not in ("cosine", "cosine-hlfperiod"):
        for _ in range(n):                   # (no shipped config; SequentialLR without warm-up has its own start-up quirks)
            sched.step()
        return
    warm, decay = sched._schedulers
    lr = lr_at(args, max_steps, n, lr_factor)
    sched.last_epoch = n
    if n < warmup:
        warm.last_epoch = n
    else:
        warm.last_epoch = max(warmup - 1, 0)
        decay.last_epoch = n - warmup if warmup > 0 else n - 1
        if warmup > 0:                       # what the last warm-up step left behind (SequentialLR never reads it again)
            warm._last_lr = [lr_at(args, max_steps, warmup - 1, lr_factor) for _ in sched.optimizer.param_groups]
    for g in sched.optimizer.param_groups:
        g["lr"] = lr
    cur = [lr for _ in sched.optimizer.param_groups]
    (warm if n < warmup else decay)._last_lr = list(cur)
    sched._last_lr = list(cur)
    # the members' step counters only gate a start-up branch (CosineAnnealingLR: _step_count == 1) and a warning
    for m_ in (sched, warm, decay):
        m_._step_count = max(getattr(m_, "_step_count", 1), 2)
