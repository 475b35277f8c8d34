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
    pmc, why = b.pmc_for_run(cfg, 10, "0123456789abcdef")
    assert pmc is None and "no PMC file" in why


def test_pmc_file_is_refused_for_another_build(tmp_path, monkeypatch):
    """The roofline's counters must come from the library that is loaded: a PMC file stamped with another source digest
    is refused (fields null, `pmc` says why), one with the right digest is used -- for all three render kernels."""
    import json

    b = _load_bench()
    cfg = {"workload": "w", "triangles": 3, "resolution": "8x8", "spp_per_step": 2, "steps": 1, "warmup": 0}
    rec = {"bench_config": {k: cfg[k] for k in ("workload", "triangles", "resolution", "spp_per_step")}, "steps": 1, "warmup": 0,
           "timed_launches": 2, "library_digest": "aaaaaaaaaaaaaaaa", "source": "test",
           "kernels": {"k_trace_extend": {"counters": {"SQ_INSTS_VALU": [5.0, 1.0, 2.0], "SQ_THREAD_CYCLES_VALU": [0.0, 64.0, 64.0]}},
                       "k_shade": {"counters": {"SQ_INSTS_VALU": [9.0, 9.0, 9.0]}}}}
    (tmp_path / "pmc_bench_t.json").write_text(json.dumps(rec))
    monkeypatch.setattr(b, "PMC_GLOB", str(tmp_path / "pmc_bench_*.json"))
    pmc, why = b.pmc_for_run(cfg, 2, "bbbbbbbbbbbbbbbb")
    assert pmc is None and "stale build" in why
    pmc, src = b.pmc_for_run(cfg, 2, "aaaaaaaaaaaaaaaa")
    assert pmc["k_trace_extend"]["SQ_INSTS_VALU"] == 3.0 and pmc["k_shade"]["SQ_INSTS_VALU"] == 18.0  # the LAST two dispatches
    assert "aaaaaaaaaaaaaaaa" in src and "scaled" not in src
    r = b.kernel_rates(pmc["k_trace_extend"], 1.0, 2, 0.0)
    assert r["lanes_per_instr"] == pytest.approx(128.0 / 3.0 / 64.0) and r["effective"] == pytest.approx(r["issue_frac"] * r["lanes_per_instr"])
    pmc, src = b.pmc_for_run(cfg, 4, "aaaaaaaaaaaaaaaa")  # another launch count: per-launch averages, said so
    assert pmc["k_trace_extend"]["SQ_INSTS_VALU"] == 6.0 and "scaled" in src
    # r05: a file that carries the digest of the render kernels' instruction streams belongs to that DEVICE CODE: it survives a
    # host-only rebuild (another library digest) and is refused for other kernels whatever the library digest says
    rec["kernel_digest"] = "kkkkkkkkkkkkkkkk"
    (tmp_path / "pmc_bench_t.json").write_text(json.dumps(rec))
    pmc, src = b.pmc_for_run(cfg, 2, "bbbbbbbbbbbbbbbb", "kkkkkkkkkkkkkkkk")
    assert pmc is not None and "kkkkkkkkkkkkkkkk" in src
    pmc, why = b.pmc_for_run(cfg, 2, "aaaaaaaaaaaaaaaa", "xxxxxxxxxxxxxxxx")
    assert pmc is None and "stale build" in why


def test_committed_counter_files_belong_to_the_library_in_the_tree():
    """profiles/pmc_bench_*.json are what the driver's bench run reads for `roofline`'s counter fields: a change of any csrc
    source or of the header re-stamps the library, so the evidence run (scripts/final_evidence_run.sh) must follow it -- this
    fails here, on CPU, when it has not."""
    import glob
    import json

    import gpuspectral_amd as g

    if os.environ.get("GSP_LIB_PATH"):
        pytest.skip("GSP_LIB_PATH points at a build variant")
    digest = g.pt.build_info()["digest"]
    vm = json.load(open(os.path.join(os.path.dirname(g.lib_path()), "valu_mix.json")))
    assert vm["library_digest"] == digest, "lib/valu_mix.json is not of this build (make -C gpuspectral_amd/csrc)"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_bench_*.json")))
    assert files
    for f in files:
        rec = json.load(open(f))
        # r05: the counters belong to the DEVICE CODE of the three render kernels (kernel_digest = sha256 of their instruction
        # streams, scripts/valu_mix.py), so a host-only change of the library does not invalidate them
        assert rec.get("kernel_digest") == vm["kernel_digest"] or rec.get("library_digest") == digest, \
            "%s was collected on other device code: rerun scripts/final_evidence_run.sh" % os.path.basename(f)


def test_gpus_defaults_to_world_size(monkeypatch, capsys):
    """`torchrun --nproc-per-node N bench.py` without --gpus runs with N ranks: no mismatch exit."""
    b = _load_bench()
    monkeypatch.setenv("WORLD_SIZE", "3")
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    args = b.parse()
    assert args.gpus is None


def test_nccl_ranks_beyond_the_visible_gpus_are_refused(monkeypatch, capsys):
    """`--gpus N` over the nccl backend with fewer than N visible GPUs exits with 2 before torch, a
    process group or a device context exists (RCCL wants one GPU per rank; a hung init would cost the driver its limit)."""
    import torch

    b = _load_bench()
    n = torch.cuda.device_count() + 2
    monkeypatch.setenv("WORLD_SIZE", str(n))
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", str(n)])
    with pytest.raises(SystemExit) as e:
        b.main()
    assert e.value.code == 2
    assert "visible GPUs" in capsys.readouterr().err


def test_a_failing_rank_exits_with_its_rank_in_the_message():
    """An exception in a rank is printed with the rank and ends the process with status 1 (os._exit, no re-exec), so the
    launcher tears the job down instead of leaving the other ranks in a barrier."""
    code = ("import sys, os; sys.argv = ['bench.py', '--scene', 'interior', '--tris', '-5']; os.environ['RANK'] = '3'; os.environ['WORLD_SIZE'] = '1';"
            "import importlib.util as u; s = u.spec_from_file_location('b', %r); m = u.module_from_spec(s); s.loader.exec_module(m);"
            "m.parse = lambda: (_ for _ in ()).throw(RuntimeError('boom')); m.main()" % os.path.join(ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 1
    assert "rank 3 of 1 failed: RuntimeError: boom" in r.stderr
