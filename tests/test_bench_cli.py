"""bench.py's command line where no GPU is needed: what it refuses."""
import os
import subprocess
import sys

from conftest import ROOT


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_n_without_n_visible_gpus_exits_non_zero_and_prints_no_line():
    """`python bench.py --gpus 8` on a box with fewer GPUs (here: none) must not print a dp1 number: rc 2, a message, no JSON.
    Decided before anything touches the GPU (torch.cuda.device_count() does not initialise it), so the same code path runs on the GPU box."""
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1 if n else 2)], env=_clean_env(), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 2, (r.returncode, r.stderr[-1000:])
    assert "GPU(s) visible" in r.stderr and "{" not in r.stdout


def test_child_processes_do_not_inherit_the_launcher_environment():
    sys.path.insert(0, ROOT)
    import bench
    keep = dict(os.environ)
    try:
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511", TORCHELASTIC_RUN_ID="x", PAPR_DIST_SINGLE="1", HOME_KEEP="1")
        env = bench.child_env()
    finally:
        os.environ.clear()
        os.environ.update(keep)
    assert not [k for k in env if k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PAPR_DIST_SINGLE") or k.startswith("TORCHELASTIC_")]
    assert env.get("HOME_KEEP") == "1"


def test_stray_world_size_without_a_launcher_is_one_rank():
    """WORLD_SIZE=8 exported by a scheduler with no RANK / MASTER_ADDR is not a launch: one rank, no group, a warning."""
    code = ("import sys, warnings; sys.path.insert(0, %r)\n"
            "from papr_amd import dist\n"
            "with warnings.catch_warnings(record=True) as w:\n"
            "    warnings.simplefilter('always'); n = dist.init_from_env('cpu')\n"
            "assert n == 1 and dist.world_size() == 1 and not dist.active() and any('WORLD_SIZE' in str(x.message) for x in w)\n" % ROOT)
    env = _clean_env()
    env["WORLD_SIZE"] = "8"
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
