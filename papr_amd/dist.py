"""Ray-sharded data parallelism: one process per GPU, RCCL over xGMI via torch.distributed.

The reference is single-process; this is new functionality whose
oracle is "the reference on one device with batch_size = world_size" (section 8e, golden G9).
Every rank renders its own patch with a full replica of the point cloud and networks; one bucketed
all-reduce per step averages all gradients (21.9 MB at P = 10k: a single ring step over xGMI costs
~0.25 ms against a multi-ms step, so one flat bucket is enough and nothing is overlapped); the three
per-point tensors are re-broadcast from rank 0 after every prune / add so that all ranks take the
decisions of rank 0's numpy RNG stream.
"""
import os

import torch
import torch.distributed as td


def launched():
    """True under a launcher (torch.distributed.run exports RANK, WORLD_SIZE and MASTER_ADDR for every rank it starts, one rank included)."""
    return all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR"))


def init_from_env(device=None):
    """Initialise the default process group when started by torch.distributed.run -- with ONE rank as well, so that a
    `--nproc-per-node 1` launch walks the same code as an 8-rank one (a plain `python bench.py` forms no group).  Returns the world size."""
    if td.is_initialized():
        return td.get_world_size()
    if not launched():
        # no launcher, no group, ONE rank -- whatever a stray WORLD_SIZE export (a scheduler's, say) claims: a caller that multiplied its
        # throughput by it, or switched to per-rank sampling with no collectives behind it, would be silently wrong
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            import warnings
            warnings.warn("papr_amd.dist: WORLD_SIZE=%s without RANK / MASTER_ADDR: not a launcher environment, running as one rank" % os.environ["WORLD_SIZE"])
        return 1
    world = int(os.environ["WORLD_SIZE"])
    use_gpu = torch.cuda.is_available() and (device is None or torch.device(device).type == "cuda")
    # PAPR_DIST_BACKEND=gloo: several ranks on ONE device (tests on a 1-GPU box: RCCL refuses two ranks per GPU); the
    # collectives then bounce device tensors through the host (_all_reduce_sum / _broadcast below)
    backend = os.environ.get("PAPR_DIST_BACKEND", "nccl" if use_gpu else "gloo")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if backend == "nccl":
        # RCCL: one rank per GPU, or the communicator errors / hangs much later -- fail here instead
        if not use_gpu:
            raise RuntimeError("PAPR_DIST_BACKEND=nccl needs a GPU")
        if local >= torch.cuda.device_count():
            raise RuntimeError("LOCAL_RANK %d but only %d GPU(s) visible: RCCL needs one GPU per rank" % (local, torch.cuda.device_count()))
        torch.cuda.set_device(local)
        td.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    elif use_gpu:
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))       # (gloo: several ranks may share a device)
        td.init_process_group(backend=backend)
    else:
        td.init_process_group(backend=backend)
    return world


def world_size():
    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


def active():
    """Do the collectives run?  With more than one rank, always.  PAPR_DIST_SINGLE=1 keeps them on in a one-rank group as well (every
    collective is then the identity): the way a 1-GPU box executes the RCCL calls of this module (tests/test_hip_rccl.py)."""
    if not (td.is_available() and td.is_initialized()):
        return False
    return td.get_world_size() > 1 or os.environ.get("PAPR_DIST_SINGLE", "0") == "1"


def rank():
    return td.get_rank() if td.is_available() and td.is_initialized() else 0


def _host_bounce(t):
    return t.is_cuda and td.get_backend() != "nccl"


def _all_reduce_sum(t):
    if _host_bounce(t):
        h = t.cpu()
        td.all_reduce(h, op=td.ReduceOp.SUM)
        t.copy_(h)
    else:
        td.all_reduce(t, op=td.ReduceOp.SUM)


def _broadcast(t, src):
    if _host_bounce(t):
        h = t.cpu()
        td.broadcast(h, src=src)
        t.copy_(h)
    else:
        td.broadcast(t, src=src)


def _dense(t):
    """Every element of t's memory span is an element of t, once (contiguous in SOME dimension order)."""
    n, expect = t.numel(), 1
    if n == 0:
        return True
    for size, stride in sorted(((sz, st) for sz, st in zip(t.size(), t.stride()) if sz != 1), key=lambda x: x[1]):
        if stride != expect:
            return False
        expect *= size
    return expect == n


def average_gradients(params):
    """One flat bucket: g <- mean over ranks of g, for every parameter that has a gradient.

    Parameters whose gradient is None on this rank (e.g. no ray touched them) contribute zeros, so
    all ranks always reduce the same layout."""
    if not active():
        return 0
    ws = world_size()
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    # the bucket holds every gradient in the MEMORY order of its parameter (the U-Net's channels-last weights included: autograd's
    # layout contract gives p.grad the strides of p), so that a view of the bucket with p's strides IS the averaged gradient
    def in_memory_order(p):
        g = p.grad if p.grad is not None else torch.zeros_like(p, memory_format=torch.preserve_format)
        if g.stride() != p.stride():
            g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
        return g.as_strided((g.numel(),), (1,)) if _dense(g) else g.contiguous().reshape(-1)

    flat = torch.cat([in_memory_order(p) for p in params])
    if td.get_backend() == "nccl":                   # RCCL averages in the collective; gloo has no AVG
        td.all_reduce(flat, op=td.ReduceOp.AVG)
    else:
        _all_reduce_sum(flat)
        flat.div_(ws)
    # the averaged gradients stay where they are: every p.grad becomes a view of the bucket (67 copy-back launches per step
    # otherwise; the optimizers' zero_grad drops the views before the next backward pass)
    off = 0
    for p in params:
        n = p.numel()
        if _dense(p):
            p.grad = flat.as_strided(p.size(), p.stride(), off)
        else:
            p.grad = torch.empty_like(p).copy_(flat[off:off + n].view_as(p))
        off += n
    return flat.numel()


def broadcast_point_cloud(tensors, src=0):
    """Broadcast a list of per-point tensors whose first dimension may differ across ranks.

    Returns new tensors (same dtype/device as the inputs) holding rank `src`'s values."""
    if not active():
        return tensors
    dev = tensors[0].device
    n = torch.tensor([tensors[0].shape[0]], device=dev, dtype=torch.int64)
    _broadcast(n, src)
    out = []
    for t in tensors:
        buf = t.detach().clone() if (rank() == src and t.shape[0] == int(n)) else \
            torch.empty((int(n),) + tuple(t.shape[1:]), device=dev, dtype=t.dtype)
        _broadcast(buf, src)
        out.append(buf)
    return out


_param_epoch = 0


def param_epoch():
    """Counts the writes to parameters that torch's version counters do not see (`.data` / raw-pointer writers: the broadcast below, papr_adam_step):
    part of the key of every cache derived from parameter values (model.ProximityAttentionParams.kernel_weights)."""
    return _param_epoch


def bump_param_epoch():
    global _param_epoch
    _param_epoch += 1


def broadcast_module_state(module, src=0):
    if not active():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        _broadcast(t.data, src)
    bump_param_epoch()
