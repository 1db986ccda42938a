"""Ray-sharded data parallelism on CPU (gloo, world_size 2): the collective plumbing of
papr_amd.dist / PAPR.step / prune / add, and the DP oracle of SURVEY.md section 8e (an N-image batch
equals the mean of N single-image gradients; reference golden G9)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as td
import torch.multiprocessing as mp

from conftest import ROOT, GOLDEN, case_cfg, golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    for p in (ROOT, GOLDEN, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(2)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = globals()[fn](rank, world)
        torch.save(res, os.path.join(out_dir, "r%d.pt" % rank))
    finally:
        td.destroy_process_group()


def run2(fn, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), fn, str(tmp_path)), nprocs=2, join=True)
    return [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(2)]


def _model(seed=1):
    from papr_amd import get_model
    torch.manual_seed(seed)
    np.random.seed(seed)
    return get_model(case_cfg("chair1k"), device="cpu")


# ---------------------------------------------------------------------------------------------
def w_step_averages(rank, world):
    from papr_amd import dist as pdist
    m = _model(seed=1 + rank)                     # replicas start different ...
    pdist.broadcast_module_state(m)               # ... and take rank 0's state, channels-last U-Net weights included
    g = torch.Generator().manual_seed(100 + rank)
    for p in m.parameters():
        if p.requires_grad:
            p.grad = torch.randn(p.shape, generator=g)
    if rank == 1:
        m.points.grad = None                      # a rank whose rays touched nothing still takes part
    grads = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters() if p.requires_grad}
    m.step(0)
    return {"grads": grads, "after": {n: p.detach().clone() for n, p in m.named_parameters()},
            "avg": {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None},
            # the averaged gradient keeps its parameter's memory format (the fused optimizers refuse anything else)
            "layout_kept": all(p.grad.stride() == p.stride() for p in m.parameters() if p.grad is not None),
            "channels_last_weights": sum(1 for p in m.parameters() if p.dim() == 4 and not p.is_contiguous())}


def test_step_allreduces_mean_gradient_and_keeps_replicas_identical(tmp_path):
    r0, r1 = run2("w_step_averages", tmp_path)
    assert r0["layout_kept"] and r1["layout_kept"] and r0["channels_last_weights"] >= 5
    for n in r0["grads"]:
        mean = 0.5 * (r0["grads"][n] + r1["grads"][n])
        assert torch.allclose(r0["avg"][n], mean, atol=1e-7), n
        assert torch.equal(r0["avg"][n], r1["avg"][n]), n
    for n in r0["after"]:
        assert torch.equal(r0["after"][n], r1["after"][n]), n
    # and it is what a single process gets from the mean gradient
    m = _model()
    for n, p in m.named_parameters():
        if p.requires_grad:
            p.grad = 0.5 * (r0["grads"][n] + r1["grads"][n])
    m.step(0)
    for n, p in m.named_parameters():
        assert torch.allclose(p.detach(), r0["after"][n], atol=1e-8), n


def w_prune_add(rank, world):
    m = _model()
    with torch.no_grad():
        m.points_influ_scores[: 200 + 100 * rank] = 1.0      # ranks disagree on purpose
        m.points += rank                                       # and have drifted apart
    m.clear_optimizer(); m.clear_scheduler()
    n = int(m.prune_points(0.0))
    m.init_optimizers(3)
    np.random.seed(50 + rank)                                  # rank-dependent RNG streams, as in training
    added = m.add_points(40)
    return {"pruned": n, "added": added, "points": m.points.detach().clone(), "influ": m.points_influ_scores.detach().clone(),
            "feats": m.pc_feats.detach().clone(), "is_param": isinstance(m.points, torch.nn.Parameter)}


def test_prune_and_add_follow_rank0(tmp_path):
    r0, r1 = run2("w_prune_add", tmp_path)
    assert r0["points"].shape == (240, 3) and r0["is_param"] and r1["is_param"]
    for key in ("points", "influ", "feats"):
        assert torch.equal(r0[key], r1[key]), key
    assert r0["added"] == 40


def w_oracle_dp(rank, world):
    """Each rank: oracle gradients of ITS image; averaged through papr_amd.dist."""
    from formula import formula_fill, synth_rays
    from oracle import papr_oracle as O
    from oracle.state import empty_state
    from papr_amd import dist as pdist
    g5, g9 = golden("g567_chair1k.npz"), golden("g9_dp.npz")
    cfg = case_cfg("chair1k")
    st = formula_fill(empty_state(cfg, 1000))
    st["points"] = torch.from_numpy(g5["points"]).clone()
    st = O.trainable_state(st, cfg)
    ro, rd, _ = synth_rays(2, 16, 16, seed=13)
    tgt = torch.from_numpy(g9["target"])
    sl = slice(rank, rank + 1)
    rgb = O.render(st, cfg, ro[sl], rd[sl])["rgb"]
    torch.mean((rgb - tgt[sl]) ** 2).backward()
    params = [t for t in st.values() if t.requires_grad]
    n = pdist.average_gradients(params)
    return {"n": n, "points": st["points"].grad.clone(), "influ": st["points_influ_scores"].grad.clone(),
            "outc": st["renderer.outc.conv.bias"].grad.clone(), "wq": st["proximity_attn.attention_layer.w_q.bias"].grad.clone()}


def test_two_rank_average_equals_reference_two_image_batch(tmp_path):
    g9 = golden("g9_dp.npz")
    r0, r1 = run2("w_oracle_dp", tmp_path)
    assert r0["n"] == 1000 * 68 + 1139288 + 3643395          # one flat bucket: P*(3+1+64) + attn + U-Net floats
    for key, ref in (("points", "both/points"), ("influ", "both/influ"), ("outc", "both/outc_bias"), ("wq", "both/wq_bias")):
        want = g9[ref]
        assert torch.equal(r0[key], r1[key])
        np.testing.assert_allclose(r0[key].numpy(), want, rtol=0, atol=1e-5 * np.abs(want).max() + 1e-9, err_msg=key)


# ---------------------------------------------------------------------------------------------
_ONE_RANK = r'''
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as td
from papr_amd import dist as pdist
assert pdist.launched() and not td.is_initialized()
assert pdist.init_from_env("cpu") == 1 and td.is_initialized() and td.get_backend() == "gloo" and td.get_world_size() == 1
assert pdist.active() == (os.environ.get("PAPR_DIST_SINGLE") == "1")
g = torch.Generator().manual_seed(0)
ps = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in [(7, 3), (7, 1), (8, 4, 3, 3), (5,)]]
ps[2].data = ps[2].data.contiguous(memory_format=torch.channels_last)
for p in ps:
    p.grad = torch.randn(p.shape, generator=g).contiguous(memory_format=torch.channels_last if p.dim() == 4 else torch.contiguous_format)
ps[1].grad = None
before = [None if p.grad is None else p.grad.clone() for p in ps]
n = pdist.average_gradients(ps)
if pdist.active():
    assert n == sum(p.numel() for p in ps)
    base = ps[0].grad.untyped_storage().data_ptr()
    for p, b in zip(ps, before):
        assert p.grad.untyped_storage().data_ptr() == base and p.grad.stride() == p.stride()
        assert torch.equal(p.grad, b if b is not None else torch.zeros_like(p))
    out = pdist.broadcast_point_cloud([ps[0].data, ps[1].data])
    assert torch.equal(out[0], ps[0].data) and out[0] is not ps[0].data
else:
    assert n == 0 and ps[1].grad is None
td.destroy_process_group()
print("ok")
'''


@pytest.mark.parametrize("single", ["1", "0"])
def test_a_one_rank_launch_forms_a_group_and_runs_the_collectives_only_on_request(single):
    """`torch.distributed.run --nproc-per-node 1` exports RANK / WORLD_SIZE = 1 / MASTER_*: init_from_env forms the group (gloo here, RCCL
    on a GPU: tests/test_hip_rccl.py); the collectives run in it with PAPR_DIST_SINGLE=1 and are short-cut without."""
    import subprocess
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), PAPR_DIST_SINGLE=single)
    r = subprocess.run([sys.executable, "-c", _ONE_RANK % ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout + r.stderr)[-3000:]
