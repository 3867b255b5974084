"""bench.py's launch contract: `python bench.py --gpus N` without a rank environment starts N ranks itself
(torch.distributed.run, one per GPU, rendezvous on 127.0.0.1) and forwards its own arguments; inside a rank
environment it does not re-launch."""
import argparse
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("i2s_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_self_launch_command(monkeypatch):
    b = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0

    monkeypatch.setattr(b.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    assert b.self_launch(argparse.Namespace(gpus=4)) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_relaunches_only_without_rank_env(monkeypatch):
    b = _bench()
    calls = []
    monkeypatch.setattr(b, "self_launch", lambda args: calls.append(args.gpus) or 0)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    try:
        b.main()
    except SystemExit as e:
        assert e.code == 0
    assert calls == [2]


def test_kernels_sha_tracks_sources():
    b = _bench()
    assert len(b.kernels_sha()) == 16
    t, t7, src = b.measured_traffic()
    # a traffic figure is only reported for the kernels it was measured on
    assert (t is None) == (src.get("kernels_sha_of_counters") != src.get("kernels_sha_now") or src.get("file") is None)
