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


def init_from_env(device=None):
    """Initialise the default process group when launched by torch.distributed.run (WORLD_SIZE > 1)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 or td.is_initialized():
        return world
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


def average_gradients(params):
    """One flat bucket: g <- mean over ranks of g, for every parameter that has a gradient.

    Parameters whose gradient is None on this rank (e.g. no ray touched them) contribute zeros, so
    all ranks always reduce the same layout."""
    ws = world_size()
    if ws == 1:
        return 0
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    if td.get_backend() == "nccl":                   # RCCL averages in the collective; gloo has no AVG
        td.all_reduce(flat, op=td.ReduceOp.AVG)
    else:
        _all_reduce_sum(flat)
        flat.div_(ws)
    # the averaged gradients stay where they are: every p.grad becomes a view of the bucket (67 copy-back launches per step
    # otherwise; the optimizers' zero_grad drops the views before the next backward pass)
    # (a parameter in another memory format -- the U-Net's channels-last weights -- gets a gradient with ITS strides, as
    # autograd's layout contract has it: the fused / foreach optimizers walk parameter and gradient memory side by side)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        p.grad = g if g.stride() == p.stride() else torch.empty_like(p).copy_(g)
        off += n
    return flat.numel()


def broadcast_point_cloud(tensors, src=0):
    """Broadcast a list of per-point tensors whose first dimension may differ across ranks.

    Returns new tensors (same dtype/device as the inputs) holding rank `src`'s values."""
    if world_size() == 1:
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


def broadcast_module_state(module, src=0):
    if world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        _broadcast(t.data, src)
