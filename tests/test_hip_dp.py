"""Ray-sharded data parallelism THROUGH the HIP path (BASELINE.json configs[4]) on a 1-GPU box:
two fresh child processes, both on cuda:0, process group gloo (RCCL refuses two ranks per device), each rendering its own
image through the real kernels; `PAPR.step()` averages the gradients.  Oracle: the reference with `batch_size = 2` on one
device (golden G9, tests/golden/g9_dp.npz)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden, grad_check

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_on_one_gpu_match_the_reference_two_image_batch(tmp_path):
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PAPR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_hip_worker.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, outs[rank][-4000:])
    r0, r1 = (torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(2))
    g9 = golden("g9_dp.npz")
    assert r0["lib"].endswith("libpapr_hip.so") and r0["abi"] == r1["abi"]
    # each rank's loss is the reference's single-image loss of ITS image
    assert abs(r0["loss"] - float(g9["img0/loss"])) < 2e-6 and abs(r1["loss"] - float(g9["img1/loss"])) < 2e-6
    # averaged gradients: identical on both ranks, == mean of the local ones, == the reference's 2-image batch
    assert set(r0["avg"]) == set(r1["avg"]) and len(r0["avg"]) > 40
    for n in r0["avg"]:
        assert torch.equal(r0["avg"][n], r1["avg"][n]), n
        l0 = r0["local"].get(n, torch.zeros_like(r0["avg"][n]))
        l1 = r1["local"].get(n, torch.zeros_like(r0["avg"][n]))
        assert torch.allclose(r0["avg"][n], 0.5 * (l0 + l1), rtol=0, atol=1e-6 * float(r0["avg"][n].abs().max()) + 1e-12), n
    for name, key in (("points", "points"), ("points_influ_scores", "influ"),
                      ("proximity_attn.attention_layer.w_q.bias", "wq_bias"), ("renderer.outc.conv.bias", "outc_bias")):
        ref = g9["both/" + key]
        got = r0["avg"][name].numpy()
        grad_check(got, ref, name)
    names = [str(x) for x in g9["both/names"]]
    for i, n in enumerate(names):                      # gradient norms of every parameter of the reference's batch-2 step
        ref_norm = g9["both/stats"][i][2]
        if n in r0["avg"]:
            assert abs(r0["avg"][n].double().norm().item() - ref_norm) <= 1e-3 * ref_norm + 1e-10, n
    # replicas stay bit-identical through the optimizer step, the prune / add round and the step after it
    for n in r0["after"]:
        assert torch.equal(r0["after"][n], r1["after"][n]), n
    assert r0["pruned"] == 700 and r1["pruned"] == 600           # each rank's own count; rank 0's cloud wins
    assert r0["added"] == 40 and r0["points2"].shape == (340, 3)
    for key in ("points2", "feats2", "influ2"):
        assert torch.equal(r0[key], r1[key]), key
