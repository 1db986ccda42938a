"""One launch for the optimizer step of all of PAPR's Adam instances (papr_adam_step, csrc/adam.hip).

Counterpart of the loop `for opt in optimizers: scaler.step(opt)` of the reference's PAPR.step (models/model.py:439-460) for
the case it is in by default on the device: GradScaler disabled (use_amp false), fp32 parameters, torch.optim.Adam without
amsgrad / maximize.  The state lives in the torch optimizers (`exp_avg`, `exp_avg_sq`, `step`, created here exactly like
torch's fused implementation creates them), so `optimizers.pth` checkpoints and `load_state_dict` are unaffected.
"""
import ctypes as C

import torch

from . import hip


class _Tensor(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("step", C.c_void_p), ("n", C.c_int64),
                ("group", C.c_int32), ("pad_", C.c_int32)]


class _Group(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double),
                ("found_slot", C.c_int32), ("pad_", C.c_int32)]


MAX_GROUPS = 8


def supported(optimizers):
    """All of them torch.optim.Adam in a configuration the kernel implements, on fp32 device parameters."""
    n_groups = 0
    for opt in optimizers:
        if type(opt) is not torch.optim.Adam:
            return False
        for g in opt.param_groups:
            n_groups += 1
            if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable") or isinstance(g["lr"], torch.Tensor):
                return False
            for p in g["params"]:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous(memory_format=torch.contiguous_format) and not p.is_contiguous(memory_format=torch.channels_last):
                    return False
    return 0 < n_groups <= MAX_GROUPS


def _adopt_state(st, p):
    """State that came from somewhere else (an `optimizers.pth` written by torch's non-fused Adam: a CPU `step`; moments saved
    contiguous and loaded next to a channels-last parameter): bring it into the form the kernel reads -- the step counter a
    float32 scalar on the parameter's device, element i of each moment at element i of the parameter in memory."""
    stp = st["step"]
    if not (torch.is_tensor(stp) and stp.is_cuda and stp.dtype == torch.float32 and stp.device == p.device):
        st["step"] = torch.as_tensor(float(stp), dtype=torch.float32, device=p.device)
    for name in ("exp_avg", "exp_avg_sq"):
        m = st[name]
        if m.device != p.device or m.dtype != torch.float32 or m.stride() != p.stride():
            st[name] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(m)


def scaler_supported(scaler):
    """step_scaled keeps torch.amp.GradScaler's per-optimizer records in step through the fields GradScaler.step / update themselves use
    (torch/amp/grad_scaler.py).  A scaler of another make, or a torch that renamed them: PAPR.step takes torch's own route instead."""
    try:
        from torch.amp.grad_scaler import OptState  # noqa: F401
    except Exception:
        return False
    if not (type(scaler).__module__.startswith("torch.") and hasattr(scaler, "_check_scale_growth_tracker") and hasattr(scaler, "_per_optimizer_states")
            and hasattr(OptState, "STEPPED") and hasattr(OptState, "UNSCALED") and hasattr(OptState, "READY")):
        return False
    # ... and the LAYOUT of a per-optimizer record, which update() reads: a fresh entry of the scaler's own defaultdict must carry the two keys
    # step_scaled writes (probed on a throw-away key, removed again)
    probe = object()
    try:
        rec = scaler._per_optimizer_states[id(probe)]
        ok = isinstance(rec, dict) and rec.get("stage") is OptState.READY and isinstance(rec.get("found_inf_per_device"), dict)
    except Exception:
        ok = False
    finally:
        try:
            scaler._per_optimizer_states.pop(id(probe), None)
        except Exception:
            ok = False
    return ok


def step_scaled(optimizers, scaler):
    """`for opt in optimizers: scaler.step(opt)` (reference models/model.py:439-442) under a live torch.amp.GradScaler, as the same launches as
    step() plus one pass that looks for non-finite gradients: the kernel divides every gradient by the scaler's scale as it reads it, and an optimizer
    with a non-finite gradient is skipped WHOLE (parameters, moments, step counters) while the others step -- torch's per-optimizer scaler.step
    contract, what the reference's loop does.  The scaler is told what it would have found itself -- found_inf per optimizer,
    stage STEPPED -- so that its update() halves / grows the scale exactly as after its own step() (torch/amp/grad_scaler.py: step, update); the
    scale never leaves the device and nothing synchronises."""
    from torch.amp.grad_scaler import OptState
    scale, _ = scaler._check_scale_growth_tracker("step")
    for opt in optimizers:
        st = scaler._per_optimizer_states[id(opt)]
        if st["stage"] is OptState.STEPPED:
            raise RuntimeError("step() has already been called since the last update().")
        if st["stage"] is OptState.UNSCALED:
            raise RuntimeError("papr_amd: gradients already unscaled by scaler.unscale_(): step through the scaler itself")
    if len(optimizers) > MAX_GROUPS:
        raise RuntimeError("papr_amd: %d optimizers (at most %d)" % (len(optimizers), MAX_GROUPS))
    found_inf = torch.empty(MAX_GROUPS + 1, dtype=torch.float32, device=scale.device)   # one slot per optimizer (each is skipped on its own overflow only) + "any"
    step(optimizers, grad_scale=scale, found_inf=found_inf)
    # update() adds up every found_inf it is shown (one small launch per optimizer beyond the first) and only asks whether the sum is positive: it is
    # shown the kernel's own "any slot" flag once, under the first optimizer
    for i, opt in enumerate(optimizers):
        st = scaler._per_optimizer_states[id(opt)]
        st["found_inf_per_device"] = {scale.device: found_inf[MAX_GROUPS:MAX_GROUPS + 1]} if i == 0 else {}
        st["stage"] = OptState.STEPPED
    return found_inf


def step(optimizers, grad_scale=None, found_inf=None):
    """optimizer.step() of every optimizer in `optimizers` (parameters without a gradient are skipped, like torch does)."""
    tensors, groups = [], []
    for oi, opt in enumerate(optimizers):
        for g in opt.param_groups:
            gi = len(groups)
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = opt.state[p]
                if len(st) == 0:                                     # (torch/optim/adam.py: _init_group, fused flavour)
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                else:
                    _adopt_state(st, p)
                grad = p.grad
                if grad.stride() != p.stride():                      # element i of the gradient must be element i of the parameter in memory
                    grad = torch.empty_like(p, memory_format=torch.preserve_format).copy_(grad)
                tensors.append(_Tensor(p.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(),
                                       p.numel(), gi, 0))
                tensors[-1]._keep = grad
            b1, b2 = g["betas"]
            groups.append(_Group(float(g["lr"]), b1, b2, g["eps"], g["weight_decay"], oi, 0))
    if not tensors:
        if found_inf is not None:
            found_inf.zero_()
        return
    arr_t = (_Tensor * len(tensors))(*tensors)
    arr_g = (_Group * len(groups))(*groups)
    if grad_scale is not None:
        hip.check(hip.lib().papr_adam_step_scaled(arr_t, len(tensors), arr_g, len(groups), hip.ptr(grad_scale), hip.ptr(found_inf), hip.stream_ptr()), "papr_adam_step_scaled")
    else:
        hip.check(hip.lib().papr_adam_step(arr_t, len(tensors), arr_g, len(groups), hip.stream_ptr()), "papr_adam_step")
    from . import dist as pdist
    pdist.bump_param_epoch()                          # (the kernel writes the parameters through raw pointers: no version counter moves)
