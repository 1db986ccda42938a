"""The RCCL branch of papr_amd/dist.py on the one GPU a test box has (BASELINE.json configs[4]).

RCCL refuses two ranks per device, so the N > 1 arithmetic is covered by tests/test_hip_dp.py (gloo, two ranks on one GPU) and
tests/test_dp_gloo.py (CPU).  What those cannot reach is the transport the metric names: `init_process_group("nccl", device_id=...)`,
`all_reduce(ReduceOp.AVG)` on the device bucket, device `broadcast`, and bench.py started by `torch.distributed.run`.  Here a ONE-rank
RCCL group runs all of it (PAPR_DIST_SINGLE=1 keeps papr_amd.dist from short-cutting a world of one), in fresh child processes."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_one_rank_rccl_group_runs_every_collective_of_the_dp_path(tmp_path):
    out = os.path.join(str(tmp_path), "rccl.json")
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               PAPR_DIST_SINGLE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PAPR_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    res = json.load(open(out))
    assert res["backend"] == "nccl" and res["lib"].endswith("libpapr_hip.so")
    assert res["pruned"] == 667 and res["added"] == 20 and res["points"] == 353


def _bench(extra_env, launcher, port=None):
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--profile-steps", "0", "--no-amp-line", "--no-shipped-line",
            "--psnr-steps", "0", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PAPR_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_under_the_launcher_prints_the_same_line_as_the_plain_launch():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 ...` -- the driver's N > 1 command line with N = 1: the launcher
    starts before any GPU call, bench.py forms the RCCL group, averages the gradient bucket every step (PAPR_DIST_SINGLE=1), takes the MAX of
    the ranks' times with an all-reduce and prints ONE line.  Same training trajectory as the plain launch; step time within the box's noise."""
    plain = _bench({}, launcher=False)
    dp = _bench({"PAPR_DIST_SINGLE": "1"}, launcher=True, port=_free_port())
    assert plain["n_gpus"] == 1 and dp["n_gpus"] == 1 and dp["config"]["parallelism"] == "dp1"
    assert dp["config"]["final_loss"] == plain["config"]["final_loss"]           # AVG over one rank is the identity: same bits
    assert dp["value"] == pytest.approx(25600 * 6 / (dp["ms_per_step"] * 6e-3), rel=1e-6)
    # one flat 21.9 MB bucket (a cat and an all-reduce) per step on top of a ~12 ms step
    assert dp["ms_per_step"] <= 1.15 * plain["ms_per_step"] + 0.3, (dp["ms_per_step"], plain["ms_per_step"])


def test_bench_gpus_2_on_a_one_gpu_box_refuses_instead_of_printing_a_dp1_line():
    """`python bench.py --gpus 2` with no launcher around it starts its own ranks (torch.distributed.run as a child process) -- and where fewer
    GPUs than ranks are visible it exits non-zero with a message and prints NO JSON line: a one-GPU number must never be read as an N-GPU one.
    A launcher whose WORLD_SIZE disagrees with --gpus is refused the same way."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one GPU visible: the self-launch would run")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PAPR_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "only 1 GPU(s) visible" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")], (r.returncode, r.stdout, r.stderr[-2000:])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and "must agree" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")], (r.returncode, r.stdout, r.stderr[-2000:])


def test_bench_line_carries_ranks_seen_and_the_logical_knn_fraction():
    j = _bench({}, launcher=False)
    assert j["ranks_seen"] == 1 and j["n_gpus"] == 1
    e = j["e2e_frac_of_knn_hbm_roofline"]
    assert e["logical"] is True and e["bytes_per_ray"] == 12 * 10000 + 12 + 4 * 20
    assert e["frac"] == pytest.approx(j["value"] * e["bytes_per_ray"] / 8e12, rel=1e-9)


def test_bench_gpus_2_walks_the_world_2_branch_on_one_gpu():
    """`python bench.py --gpus 2` end to end on the one GPU of a test box (the first SCALE run must not be this code's first run):
    self_launch starts two ranks through torch.distributed.run, PAPR_BENCH_SHARE_GPU=1 (a test-only override of the visible-GPU check) lets them share
    cuda:0 over gloo; each rank renders its own patches, PAPR.step() averages the gradient bucket, the ranks' times meet in an all-reduce(MAX), rank 0's
    line is relayed with the launcher's exit code.  The line must say what ran: two ranks, dp2, twice one rank's rays per step over the slowest rank's time."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--profile-steps", "0", "--no-amp-line", "--no-shipped-line",
           "--psnr-steps", "0", "--no-cpu-baseline"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PAPR_BENCH_SHARE_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PAPR_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                       # ONE line: rank 0's
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["config"]["parallelism"] == "dp2" and j["scaling"] == "weak"
    assert "test_override" in j and "starting 2 ranks" in r.stderr
    R = j["config"]["global_batch_rays"] // 2
    assert R == 25600
    assert abs(j["value"] - 2 * R * 1e3 / j["ms_per_step"]) <= 1e-6 * j["value"]
    one = _bench({}, launcher=False)
    assert one["config"]["global_batch_rays"] == R and j["ms_per_step"] > one["ms_per_step"]       # (two ranks took turns on one GPU: slower per step, as it must be)
    # the exit code of a failing rank comes back through self_launch (here: a scene file that does not exist), and no line is printed
    bad = subprocess.run(cmd + ["--scene", "nerfsyn/no_such_scene.yml"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith("{")]
