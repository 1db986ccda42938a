"""The one rank of tests/test_hip_rccl.py: a fresh process started like torch.distributed.run starts a rank (RANK / WORLD_SIZE / LOCAL_RANK /
MASTER_* in the environment), WORLD_SIZE = 1, backend nccl (= RCCL on ROCm), PAPR_DIST_SINGLE=1 so that papr_amd.dist runs its collectives in
the one-rank group instead of short-cutting them.  Every line of papr_amd/dist.py that an 8-GPU launch executes runs here on device tensors;
with one rank each collective is the identity, which is what the asserts check, bit for bit.

    python tests/rccl_worker.py <out.json>
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(out_path):
    import torch.distributed as td
    from conftest import case_cfg
    from formula import formula_fill, synth_rays
    from papr_amd import adam as own_adam, dist as pdist, get_model, hip
    res = {}
    world = pdist.init_from_env("cuda")
    assert world == 1 and td.is_initialized() and td.get_backend() == "nccl" and td.get_world_size() == 1
    assert pdist.active() and pdist.rank() == 0
    dev = torch.device("cuda", 0)

    # ---- average_gradients on device tensors: ReduceOp.AVG over RCCL, gradients become views of ONE bucket --------------------------
    g = torch.Generator().manual_seed(3)
    shapes = [(1001, 3), (1001, 1), (1001, 64), (256, 117), (256,), (128, 32, 3, 3), (3, 128, 1, 1)]       # 3 x 1001 is no multiple of 4: the views
    params = []                                                                                            # behind it start at 4-byte alignment
    for i, sh in enumerate(shapes):
        p = torch.nn.Parameter(torch.randn(sh, generator=g).to(dev))
        if len(sh) == 4 and i == 5:
            p.data = p.data.contiguous(memory_format=torch.channels_last)
        p.grad = torch.randn(sh, generator=g).to(dev)
        if p.dim() == 4 and not p.is_contiguous():
            p.grad = p.grad.contiguous(memory_format=torch.channels_last)
        params.append(p)
    params[1].grad = None                                    # "no ray touched it": contributes zeros
    before = [None if p.grad is None else p.grad.clone() for p in params]
    n = pdist.average_gradients(params)
    torch.cuda.synchronize()
    assert n == sum(p.numel() for p in params)
    base = params[0].grad.untyped_storage().data_ptr()
    for p, b in zip(params, before):
        assert p.grad.untyped_storage().data_ptr() == base, "gradient is not a view of the bucket"
        assert p.grad.stride() == p.stride()
        assert torch.equal(p.grad, b if b is not None else torch.zeros_like(p)), "AVG over one rank changed a gradient"
    res["bucket_floats"] = n
    res["misaligned_views"] = sum(1 for p in params if p.grad.data_ptr() % 16 != 0)
    assert res["misaligned_views"] >= 1
    # ... and those views are what the one-launch optimizer step takes (papr_adam_step) -- against torch's Adam on copies
    ref = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in params]
    for r, p in zip(ref, params):
        r.grad = p.grad.detach().clone(memory_format=torch.preserve_format)
    opt = torch.optim.Adam(params, lr=1e-2)
    opt_ref = torch.optim.Adam(ref, lr=1e-2)
    assert own_adam.supported([opt])
    own_adam.step([opt])
    opt_ref.step()
    torch.cuda.synchronize()
    res["adam_max_diff"] = max(float((p - r).abs().max()) for p, r in zip(params, ref))
    assert res["adam_max_diff"] <= 1e-6

    # ---- broadcast_point_cloud / broadcast_module_state: device broadcasts -----------------------------------------------------------
    pts = [torch.randn((777, 3), generator=g).to(dev), torch.randn((777, 1), generator=g).to(dev), torch.randn((777, 64), generator=g).to(dev)]
    out = pdist.broadcast_point_cloud(pts)
    assert all(o is not t and torch.equal(o, t) for o, t in zip(out, pts))

    # ---- the model's own step under the group: PAPR.step() averages, prune / add broadcast ---------------------------------------------
    def one_model():
        torch.manual_seed(1)
        np.random.seed(1)
        m = get_model(case_cfg("chair1k"), device="cpu")
        formula_fill(m.state_dict())
        m = m.to(dev)
        m.clear_optimizer(); m.clear_scheduler(); m.init_optimizers(0)
        return m

    ro, rd, c2w = synth_rays(1, 16, 16, seed=0)
    ro, rd, c2w = ro.to(dev), rd.to(dev), c2w.to(dev)

    def two_steps(m):
        losses = []
        for step in range(2):
            m.clear_grad()
            loss = torch.mean((m(ro, rd, c2w, step) - 0.5) ** 2)
            loss.backward()
            m.step(step)
            losses.append(float(loss))
        return losses

    m_dp = one_model()
    pdist.broadcast_module_state(m_dp)
    l_dp = two_steps(m_dp)
    views = sum(1 for p in m_dp.parameters() if p.grad is not None and p.grad.untyped_storage().nbytes() > 4 * p.numel())
    assert views > 40, "PAPR.step() did not run the bucketed all-reduce (%d gradient views)" % views
    m_dp.clear_optimizer(); m_dp.clear_scheduler()
    with torch.no_grad():
        m_dp.points_influ_scores.zero_()
        m_dp.points_influ_scores[:333] = 1.0                 # 333 points stay: 999 floats, the bucket's views go off 16-byte alignment
    res["pruned"] = int(m_dp.prune_points(0.0))
    m_dp.init_optimizers(2)
    np.random.seed(7)
    m_dp.clear_optimizer(); m_dp.clear_scheduler()
    res["added"] = int(m_dp.add_points(20))
    m_dp.init_optimizers(2)
    l_dp += two_steps(m_dp)
    os.environ["PAPR_DIST_SINGLE"] = "0"                     # the same model and steps with the collectives short-cut
    assert not pdist.active()
    m_1 = one_model()
    l_1 = two_steps(m_1)
    m_1.clear_optimizer(); m_1.clear_scheduler()
    with torch.no_grad():
        m_1.points_influ_scores.zero_()
        m_1.points_influ_scores[:333] = 1.0
    m_1.prune_points(0.0)
    m_1.init_optimizers(2)
    np.random.seed(7)
    m_1.clear_optimizer(); m_1.clear_scheduler()
    m_1.add_points(20)
    m_1.init_optimizers(2)
    l_1 += two_steps(m_1)
    torch.cuda.synchronize()
    assert l_dp == l_1, (l_dp, l_1)
    for (na, a), (nb, b) in zip(m_dp.named_parameters(), m_1.named_parameters()):
        assert na == nb and torch.equal(a, b), na
    res.update(losses=l_dp, points=int(m_dp.points.shape[0]), lib=hip.LIB_PATH, backend=td.get_backend(),
               nccl_version=list(torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None)
    td.barrier()
    td.destroy_process_group()
    json.dump(res, open(out_path, "w"))


if __name__ == "__main__":
    main(sys.argv[1])
