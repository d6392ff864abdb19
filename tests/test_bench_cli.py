"""`python3 bench.py --gpus N` as ONE command (`torchrun --nproc-per-node N`): started alone with N > 1 the
script launches its N ranks itself as a child `python -m torch.distributed.run`, before any GPU call, and relays the JSON line.

CPU part: the command the parent builds.  GPU part: the exact command line of the driver with `--gpus 2` on a one-GPU box under the
rehearsal switch (both ranks on device 0, tiles gathered over gloo) -- the protocol only, its numbers mean nothing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_alone_with_gpus_above_one_it_starts_its_own_ranks(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the child's exit code is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"]      # the same arguments, unchanged
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_a_rank_count_that_does_not_match_the_launcher_is_refused(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "does not match WORLD_SIZE" in str(e.value.code)


@pytest.mark.gpu
def test_the_drivers_command_with_two_ranks_on_one_gpu():
    env = dict(os.environ, SDN_REHEARSE_ON_ONE_GPU="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--size", "200",
           "--no-cpu-baseline", "--no-secondary", "--min-timed-s", "0"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, out.stdout[-3000:]                  # rank 0 prints ONE line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 4 and rec["warmup"] == 1
    assert rec["value"] > 0 and rec["unit"] == "sampled-points/s" and rec["scaling"] in ("strong", "weak")
