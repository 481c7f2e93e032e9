"""CPU: bench.self_launch -- what `python bench.py --gpus N` execs when no launcher is around it, and what it refuses."""
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(**kw):
    a = types.SimpleNamespace(gpus=1, mode="retrieval", share_gpu=False)
    a.__dict__.update(kw)
    return a


@pytest.fixture
def no_launcher(monkeypatch):
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)


def test_one_gpu_and_launched_runs_do_not_exec(monkeypatch, no_launcher):
    monkeypatch.setattr(os, "execvpe", lambda *a: pytest.fail("must not exec"))
    bench.self_launch(_args(gpus=1))
    monkeypatch.setenv("WORLD_SIZE", "8")
    bench.self_launch(_args(gpus=8))  # under torch.distributed.run: nothing to do


def test_world_size_mismatch_is_an_error(monkeypatch, no_launcher):
    """A LAUNCHER (RANK / LOCAL_RANK set) that started another number of ranks than --gpus: refuse to print a line whose n_gpus is not --gpus."""
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit) as e:
        bench.self_launch(_args(gpus=8))
    assert "WORLD_SIZE=1" in str(e.value)
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.delenv("RANK")
    with pytest.raises(SystemExit) as e:
        bench.self_launch(_args(gpus=8))
    assert "WORLD_SIZE=4" in str(e.value)


def test_an_exported_world_size_of_one_without_a_launcher_self_launches(monkeypatch, no_launcher):
    """an environment that merely exports WORLD_SIZE=1 (no RANK / LOCAL_RANK: no launcher around this process) used to abort `--gpus 8`; it now starts
    the 8 ranks itself, with the stale variables removed from the launcher's environment."""
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    seen = {}

    def fake_exec(exe, cmd, env):
        seen.update(cmd=cmd, env=env)
        raise SystemExit(0)
    monkeypatch.setattr(os, "execvpe", fake_exec)
    with pytest.raises(SystemExit):
        bench.self_launch(_args(gpus=8))
    assert seen["cmd"][seen["cmd"].index("--nproc-per-node") + 1] == "8" and "WORLD_SIZE" not in seen["env"]


def test_execs_torch_distributed_run_with_n_ranks(monkeypatch, no_launcher):
    seen = {}
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "5", "--warmup", "2"])

    def fake_exec(exe, cmd, env):
        seen.update(exe=exe, cmd=cmd, env=env)
        raise SystemExit(0)
    monkeypatch.setattr(os, "execvpe", fake_exec)
    with pytest.raises(SystemExit):
        bench.self_launch(_args(gpus=8))
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_fewer_devices_than_ranks_is_refused_unless_shared(monkeypatch, no_launcher):
    monkeypatch.setattr(bench.torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.self_launch(_args(gpus=2))
    assert "only 1 HIP device" in str(e.value)
    monkeypatch.setattr(os, "execvpe", lambda *a: (_ for _ in ()).throw(SystemExit(0)))
    with pytest.raises(SystemExit) as e:
        bench.self_launch(_args(gpus=2, share_gpu=True))
    assert e.value.code == 0
