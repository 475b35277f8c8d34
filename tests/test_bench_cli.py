"""bench.py's command-line contract on a box without a GPU: `--gpus N` without a launcher starts
torch.distributed.run as a CHILD process (never replaces itself, never touches the GPU first) and returns its exit
code; a launcher that started a different number of ranks is refused."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _load_bench():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_gpus_without_launcher_spawns_torchrun_child(monkeypatch):
    b = _load_bench()
    seen = {}

    def fake_call(cmd, *a, **k):
        seen["cmd"] = cmd
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2", "--warmup", "1"])
    with pytest.raises(SystemExit) as e:
        b.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "2", "--warmup", "1"] and cmd[-7].endswith("bench.py")


def test_mismatched_world_size_is_refused(monkeypatch, capsys):
    b = _load_bench()
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3"])
    with pytest.raises(SystemExit) as e:
        b.main()
    assert e.value.code == 2
    assert "WORLD_SIZE 1" in capsys.readouterr().err


def test_pmc_file_is_refused_for_another_workload():
    b = _load_bench()
    cfg = {"workload": "something else", "triangles": 1, "resolution": "8x8", "spp_per_step": 1}
    assert b.pmc_for_run(cfg, 10) is None
