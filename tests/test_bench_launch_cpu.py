"""bench.py's multi-GPU entry: `--gpus N` must give a job of N ranks or fail - never the one-GPU number under
another label.  The decision is a pure function of the flags and the environment (no GPU call), so it is checked here."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_launch_plan_decisions():
    lp = bench.launch_plan
    assert lp(1, {}, 1) == ("run", 1)
    assert lp(1, {}, 0) == ("run", 1)                                   # (the native context fails loudly later without a GPU)
    assert lp(8, {}, 8) == ("spawn", 8)                                 # no launcher: start the ranks ourselves
    assert lp(2, {}, 8) == ("spawn", 2)
    assert lp(8, {}, 1)[0] == "fail" and "only 1 GPU" in lp(8, {}, 1)[1]
    assert lp(8, {"WORLD_SIZE": "8", "LOCAL_WORLD_SIZE": "8"}, 8) == ("run", 8)   # the driver's torch.distributed.run form
    assert lp(8, {"WORLD_SIZE": "4"}, 8)[0] == "fail"                   # flags and launcher disagree
    assert lp(1, {"WORLD_SIZE": "2"}, 8)[0] == "fail"
    assert lp(1, {"WORLD_SIZE": "1"}, 1) == ("run", 1)                  # one rank under a launcher (RHO_TTS_AMD_FORCE_DIST tests)
    assert lp(4, {"WORLD_SIZE": "4", "LOCAL_WORLD_SIZE": "4"}, 2)[0] == "fail"
    assert lp(2, {"WORLD_SIZE": "2"}, 1, backend="gloo") == ("run", 2)  # gloo rehearsal: ranks may share a GPU
    assert lp(2, {}, 1, backend="gloo") == ("spawn", 2)
    assert lp(0, {}, 8)[0] == "fail"
    assert lp(2, {"WORLD_SIZE": "x"}, 8)[0] == "fail"


def test_gpus_more_than_visible_exits_non_zero():
    """`python bench.py --gpus 2` where fewer GPUs are visible (none in the build container, one on a 1-GPU box) must not
    print a number: exit code != 0 and no JSON line."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a host with fewer than 2 visible GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_exits_non_zero():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def test_spawn_command_is_a_child_process(monkeypatch):
    """The ranks are started with subprocess (a child), on 127.0.0.1, with the same flags - and never by replacing this process."""
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(os, "execv", lambda *a, **k: (_ for _ in ()).throw(AssertionError("exec is forbidden")))
    rc = bench.spawn_ranks(8, ["--gpus", "8", "--steps", "5", "--warmup", "1"])
    assert rc == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
