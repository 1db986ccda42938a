"""One rank of the 2-rank data-parallel test on ONE GPU (started as a fresh child process by
tests/test_hip_dp.py; both ranks use cuda:0, process group gloo with device tensors bounced through the host).

    python tests/dp_hip_worker.py <out_dir>      (RANK / WORLD_SIZE / MASTER_* / PAPR_DIST_BACKEND from the environment)

Each rank renders ITS 16x16 image of the DP golden (tests/golden/g9_dp.npz: the reference with a 2-image batch,
SURVEY.md section 8e) through the real HIP forward / backward, takes PAPR.step() -- which averages the gradients over
the ranks -- and then one prune + add round with rank-dependent numpy streams.  Everything the test compares goes
into <out_dir>/r<rank>.pt.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main(out_dir):
    from conftest import case_cfg, golden
    from formula import formula_fill, synth_rays
    from papr_amd import dist as pdist, get_model, hip
    world = pdist.init_from_env("cuda")
    rank = pdist.rank()
    assert world == 2 and torch.distributed.get_backend() == "gloo"
    dev = torch.device("cuda", 0)
    g5, g9 = golden("g567_chair1k.npz"), golden("g9_dp.npz")
    torch.manual_seed(1 + rank)                       # replicas start different on purpose ...
    np.random.seed(1 + rank)
    m = get_model(case_cfg("chair1k"), device="cpu")
    if rank == 0:
        formula_fill(m.state_dict())
        with torch.no_grad():
            m.points.copy_(torch.from_numpy(g5["points"]))
    m = m.to(dev)
    m.clear_optimizer(); m.clear_scheduler(); m.init_optimizers(0)
    pdist.broadcast_module_state(m)                   # ... and take rank 0's state
    ro, rd, c2w = synth_rays(2, 16, 16, seed=13)
    tgt = torch.from_numpy(g9["target"])
    sl = slice(rank, rank + 1)
    m.clear_grad()
    rgb = m(ro[sl].to(dev), rd[sl].to(dev), c2w[sl].to(dev), 0)
    loss = torch.mean((rgb - tgt[sl].to(dev)) ** 2)
    m.scaler.scale(loss).backward()
    local = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}
    m.step(0)
    m.scaler.update()
    avg = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters() if p.grad is not None}
    after = {n: p.detach().cpu().clone() for n, p in m.named_parameters()}
    # one prune + add round under DP: the ranks disagree on the scores and on the numpy stream; rank 0 decides
    with torch.no_grad():
        m.points_influ_scores.zero_()
        m.points_influ_scores[: 300 + 100 * rank] = 1.0
    m.clear_optimizer(); m.clear_scheduler()
    pruned = int(m.prune_points(0.0))
    m.init_optimizers(1)
    np.random.seed(50 + rank)
    m.clear_optimizer(); m.clear_scheduler()
    added = int(m.add_points(40))
    m.init_optimizers(1)
    # and the replicas still step together on the new cloud
    m.clear_grad()
    rgb2 = m(ro[sl].to(dev), rd[sl].to(dev), c2w[sl].to(dev), 1)
    torch.mean((rgb2 - tgt[sl].to(dev)) ** 2).backward()
    m.step(1)
    torch.cuda.synchronize()
    torch.save({"loss": float(loss), "local": local, "avg": avg, "after": after, "pruned": pruned, "added": added,
                "points2": m.points.detach().cpu().clone(), "feats2": m.pc_feats.detach().cpu().clone(),
                "influ2": m.points_influ_scores.detach().cpu().clone(), "lib": hip.LIB_PATH,
                "abi": int(hip.lib().papr_abi_version())}, os.path.join(out_dir, "r%d.pt" % rank))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
