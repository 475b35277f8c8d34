"""Several GPUs' worth of shares on the ONE GPU of the test box: tile partition + per-share pipelines + the gather
into the first device + the scatter kernel, through the C ABI (gsp_multi_*), through the C++ host class
(MultiGpuPathTracer, via the gsp_render CLI) and through bench.py's one-process-per-rank path (gloo stands in for RCCL,
two processes share the GPU).  Any partition must reproduce the single-GPU frame bit for bit (the seed depends on the
global pixel index and the timestamp only, raygen.rgen:37) -- BASELINE config 4 minus the xGMI links."""
import json
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

from conftest import CORNELL_XML, ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shares_on_one_gpu_reproduce_the_single_gpu_frame(materials_scene, world):
    import gpuspectral_amd as g
    from gpuspectral_amd import pt

    W, H, spp = 136, 100, 6  # 5 x 4 tiles, the last column / row partial
    with g.Context(0) as ctx:
        ctx.upload_scene(materials_scene)
        ctx.frame_begin(W, H)
        ctx.render(spp=spp)
        ref = ctx.download()
        st1 = ctx.stats()
    with pt.MultiContext([0] * world) as m:
        m.upload_scene(materials_scene)
        m.frame_begin(W, H)
        m.render(spp=2)
        m.render(spp=spp - 2, first_timestamp=2)  # (calls pipeline across shares as they do on one context)
        img = m.download()
        tot, each = m.stats(per_share=True)
        assert np.array_equal(img, ref)
        for k in ("extension_rays", "shadow_rays", "shaded_vertices", "samples"):
            assert tot[k] == st1[k], k
        assert len(each) == world and sum(e["samples"] for e in each) == W * H * spp
        sizes = [e["samples"] // spp for e in each]
        assert sizes == [len(pt.tile_partition(W, H, r, world)) for r in range(world)]
        # a second frame on the same object (frame_begin re-partitions, accumulate buffers are cleared)
        m.frame_begin(W, H)
        m.render(spp=spp)
        assert np.array_equal(m.download(), ref)


def test_more_shares_than_tiles(materials_scene):
    """A 40 x 40 frame has four 32 x 32 tiles: with eight shares, four of them own no pixel at all."""
    import gpuspectral_amd as g
    from gpuspectral_amd import pt

    W, H, spp = 40, 40, 5
    with g.Context(0) as ctx:
        ctx.upload_scene(materials_scene)
        ctx.frame_begin(W, H)
        ctx.render(spp=spp)
        ref = ctx.download()
    assert sum(len(pt.tile_partition(W, H, r, 8)) == 0 for r in range(8)) == 4
    with pt.MultiContext([0] * 8) as m:
        m.upload_scene(materials_scene)
        m.frame_begin(W, H)
        m.render(spp=spp)
        assert np.array_equal(m.download(), ref)
        tot, each = m.stats(per_share=True)
        assert tot["samples"] == W * H * spp and sum(e["samples"] == 0 for e in each) == 4


def test_config4_partition_at_full_resolution():
    """BASELINE config 4 (bathroom2 stand-in, 1920x1080, tiled over 8 GPUs) with the eight shares on one GPU: the
    gathered frame equals the single-context frame.  8 spp here; the full 4096 spp of the same frame are checked
    against the oracle in tests/test_gpu_full_configs.py."""
    import gpuspectral_amd as g
    from gpuspectral_amd import pt, scenes

    sc = scenes.interior(600_000, seed=7)
    W, H, spp = 1920, 1080, 8
    with g.Context(0) as ctx:
        ctx.upload_scene(sc)
        ctx.frame_begin(W, H)
        ctx.render(spp=spp)
        ref = ctx.download()
        st1 = ctx.stats()
    with pt.MultiContext([0] * 8) as m:
        m.upload_scene(sc)
        m.frame_begin(W, H)
        m.render(spp=spp)
        img = m.download()
        tot = m.stats()
    assert zlib.crc32(img.tobytes()) == zlib.crc32(ref.tobytes())
    assert tot["extension_rays"] == st1["extension_rays"] and tot["shadow_rays"] == st1["shadow_rays"]


def test_rccl_gather_route_with_one_device(monkeypatch, materials_scene):
    """The RCCL route of gsp_multi_gather on the one GPU of the test box: GSP_MULTI_GATHER=rccl makes a single share
    build a communicator (ncclCommInitAll over one device) and send / receive its tiles to itself inside one
    ncclGroupStart / ncclGroupEnd, then scatter them: the frame must equal the single-context frame.  (With one
    share per device this is the default route; a device list with repeats cannot form a communicator.)"""
    import gpuspectral_amd as g
    from gpuspectral_amd import pt

    W, H, spp = 320, 200, 6
    with g.Context(0) as ctx:
        ctx.upload_scene(materials_scene)
        ctx.frame_begin(W, H)
        ctx.render(spp=spp)
        ref = ctx.download()
    monkeypatch.setenv("GSP_MULTI_GATHER", "rccl")
    with pt.MultiContext([0]) as m:
        assert m.gather_route()[0] == "rccl"
        m.upload_scene(materials_scene)
        m.frame_begin(W, H)
        m.render(spp=spp)
        img = m.download()
        route, n_rccl, n_copy = m.gather_route()
        assert route == "rccl" and n_rccl >= 1 and n_copy == 0
        m.render(spp=2, first_timestamp=spp)  # a second gather through the same communicator
        img2 = m.download()
        assert m.gather_route()[1] >= 2
    assert np.array_equal(img, ref)
    assert np.isfinite(img2).all()
    monkeypatch.setenv("GSP_MULTI_GATHER", "copy")
    with pt.MultiContext([0, 0]) as m:  # repeats: always the copy route
        assert m.gather_route()[0] == "copy"
    monkeypatch.delenv("GSP_MULTI_GATHER")
    with pt.MultiContext([0, 0]) as m:
        assert m.gather_route()[0] == "copy"


def test_rccl_is_resolved_at_run_time_with_a_fallback(materials_scene):
    """The C++ product carries the RCCL gather without LINKING librccl (a hard -lrccl made a single-GPU dlopen
    fail where RCCL is absent): `readelf -d` shows no librccl, the forced RCCL route above still works (dlopen found it),
    GSP_GATHER_RCCL with a repeated device is refused with a message, and GSP_GATHER_AUTO on a repeated device list takes
    the copy route and says why."""
    import gpuspectral_amd as g
    from gpuspectral_amd import abi, pt

    lib = os.path.join(ROOT, "gpuspectral_amd", "lib", "libgpuspectral_pt.so")
    out = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True).stdout
    assert "librccl" not in out, out
    with pytest.raises(g.GspError, match="one rank per device"):
        pt.MultiContext([0, 0], options=abi.CtxOptions(gather_route=abi.GATHER_RCCL))
    with pt.MultiContext([0, 0], options=abi.CtxOptions(gather_route=abi.GATHER_AUTO)) as m:
        assert m.gather_route()[0] == "copy"
        assert "repeats" in m._L.gsp_multi_last_error(m._h).decode()
    with pt.MultiContext([0], options=abi.CtxOptions(gather_route=abi.GATHER_RCCL)) as m:
        assert m.gather_route()[0] == "rccl" and "RCCL" in m._L.gsp_multi_last_error(m._h).decode()


def test_bench_one_rank_through_nccl(tmp_path):
    """bench.py with ONE rank but through torch.distributed's "nccl" backend (= RCCL): process-group creation, the
    torch-bundled HIP runtime next to the hipcc-built library, the device-to-device hand-over of the accumulate
    buffer into a torch CUDA tensor and a (one-rank) dist.gather, on every driver run -- so that the first
    execution of this path is not the 8-GPU scaling run.  Frame and ray counts equal the plain single-rank run's."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "1", "--warmup", "1", "--spp-per-step", "4", "--tris", "60000", "--width", "416", "--height", "240",
              "--no-cpu-baseline"]
    f1, f2 = str(tmp_path / "plain.npy"), str(tmp_path / "nccl.npy")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dump", f1] + common, env=env,
                        capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-3000:]
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--backend", "nccl", "--dump", f2] + common,
                        env=env, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    j1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    j2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    assert j1["config"]["collective"] is None and j2["config"]["collective"].startswith("nccl, 1 rank")
    # r06: the N = 1 line has the frame read-back inside the timed region and BASELINE's other configs behind it
    assert 0.0 < j1["config"]["download_ms"] < 1e3 * j1["ms_per_step"] and j2["config"]["download_ms"] is None
    ow = j1["config"]["other_workloads"]
    assert [w["workload"].split(":")[0] for w in ow] == ["config2", "config3", "config5"] and all(w["mrays_per_s"] > 100 and w["spp"] >= 8 for w in ow)
    assert ow[2]["max_depth"] == 32 and ow[2]["resolution"] == "4096x4096" and ow[2]["pixels"] == 4096 * 4096 // 8
    assert j1["roofline"]["bound"] in (None, "valu_issue", "l1_request", "hbm") and set(j1["cpu_baseline"] if "cpu_baseline" in j1 else ()) == set()
    assert j1["config"]["extension_rays"] == j2["config"]["extension_rays"] and j1["config"]["shadow_rays"] == j2["config"]["shadow_rays"]
    a, b = np.load(f1), np.load(f2)
    assert a.shape == (240, 416, 4) and np.array_equal(a.reshape(-1, 4), b.reshape(-1, 4))


def _read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline() == b"PF\n"
        w, h = map(int, f.readline().split())
        assert float(f.readline()) < 0
        return np.frombuffer(f.read(), np.float32).reshape(h, w, 3)[::-1]


def test_cpp_host_multi_gpu_tracer_via_cli(tmp_path):
    """The C++ host class MultiGpuPathTracer (gsp_render ... 0,0,0: three shares on device 0) against PathTracer."""
    lib = os.path.join(ROOT, "gpuspectral_amd", "lib")
    env = dict(os.environ, LD_LIBRARY_PATH=lib + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    outs = []
    for devs in ("0", "0,0,0"):
        out = str(tmp_path / ("c_%d.pfm" % len(devs)))
        r = subprocess.run([os.path.join(lib, "gsp_render"), CORNELL_XML, out, "200", "120", "5", devs], env=env,
                           capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stderr
        if devs != "0":
            assert "3 shares" in r.stdout
        outs.append(_read_pfm(out))
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("ranks", [2, 8])
def test_bench_ranks_on_one_gpu(tmp_path, ranks):
    """bench.py's N > 1 path end to end on hardware: N processes (torch.distributed.run, gloo instead of RCCL because
    the ranks share the one GPU), the real GPU renderer, C++ tile partition, gather, assembly, the per-rank record --
    against bench.py's own single-rank frame.  N = 8 is the shape of the driver's scaling run (scripts/ranks8_one_gpu.sh)."""
    env = dict(os.environ, GSP_POOL_PATHS="2000000")  # (the test binding maps it onto gsp_ctx_options: 8 pools share one GPU)
    common = ["--steps", "1", "--warmup", "1", "--spp-per-step", "4", "--tris", "60000", "--width", "416", "--height", "240",
              "--no-cpu-baseline"]
    f1, f2 = str(tmp_path / "one.npy"), str(tmp_path / "two.npy")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dump", f1] + common, env=env,
                        capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-3000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                         "--master-port", str(29613 + ranks), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--backend", "gloo", "--dump", f2] + common,
                        env=env, capture_output=True, text=True, timeout=1200)
    assert r2.returncode == 0, r2.stderr[-3000:]
    j1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    j2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == ranks
    pr = j2["config"]["per_rank"]
    assert [p["rank"] for p in pr] == list(range(ranks)) and sum(p["pixels"] for p in pr) == 416 * 240
    assert sum(p["samples"] for p in pr) == j2["config"]["spp_timed"] * 416 * 240 and all(p["elapsed_s"] > 0 and p["device_bytes"] > 0 for p in pr)
    # path segments are a property of the samples; how many of the camera-ray segments were copied from the primary-hit
    # memo instead of traced is not (a rank whose few remaining paths go to k_finish traces them from the camera)
    assert j1["config"]["path_segments"] == j2["config"]["path_segments"] and j1["config"]["shadow_rays"] == j2["config"]["shadow_rays"]
    a, b = np.load(f1), np.load(f2)
    assert a.shape == b.shape == (240, 416, 4) and np.array_equal(a, b)


_TORCH_PIECES = r"""
import sys
import numpy as np
import torch  # first, as in bench.py: the tracer's library then binds to the HIP runtime torch has loaded
assert torch.cuda.is_available()
sys.path.insert(0, %r)
import gpuspectral_amd as g
from gpuspectral_amd import multigpu, scenes

sc = scenes.cornell_materials(12)
W, H, spp, world = 200, 120, 3, 2
with g.Context(0) as ctx:
    ctx.upload_scene(sc)
    ctx.frame_begin(W, H)
    ctx.render(spp=spp)
    full = ctx.download().reshape(-1, 4)
    parts = []
    for rank in range(world):
        ids = multigpu.partition(W, H, rank, world)
        ctx.frame_begin(W, H, ids)
        ctx.render(spp=spp)
        t = torch.zeros((len(ids), 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ctx.copy_accum_to_device(t.data_ptr(), len(ids) * 16)
        assert np.array_equal(t.cpu().numpy(), ctx.download_compact())
        parts.append(t)


class FakeDist:  # rank 0's view of dist.gather: its own padded buffer + the one rank 1 would have sent
    def gather(self, buf, gather_list, dst=0):
        counts = [len(multigpu.partition(W, H, r, world)) for r in range(world)]
        for r in range(world):
            gather_list[r].zero_()
            gather_list[r][: counts[r]] = parts[r]


frame = multigpu.gather_frame(parts[0], W, H, 0, world, FakeDist())
assert frame.is_cuda and np.array_equal(frame.cpu().numpy(), full)
print("torch pieces ok")
"""


def test_device_side_gather_pieces_with_torch_on_the_gpu():
    """What the RCCL path of bench.py does on each rank besides the collective itself, in one fresh process that imports
    torch first like bench.py: the tracer's accumulate buffer copied device-to-device into a torch CUDA tensor
    (gsp_copy_accum_to_device), and rank 0's assembly of the padded per-rank buffers with index tensors on the GPU
    (multigpu.gather_frame with a stand-in for torch.distributed whose gather hands over the other rank's buffer).
    torch.cuda and the tracer's own HIP context share the device."""
    pytest.importorskip("torch")
    r = subprocess.run([sys.executable, "-c", _TORCH_PIECES % ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "torch pieces ok" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]


# ---- N > 1 PHYSICAL GPUs ---------------------------------------------------------------------------
# Every test above runs on the one GPU of the test pool.  These three need two or more devices and are skipped there (the
# skip shows up in the suite's summary): the minute a multi-GPU node runs the suite, the real RCCL group over distinct
# devices and bench.py's one-process-per-GPU path prove themselves against the single-GPU frame, bit for bit.
def _gpus():
    import gpuspectral_amd as g

    return g.device_count()


needs_two_gpus = pytest.mark.skipif("_gpus() < 2", reason="needs >= 2 physical GPUs (the test pool has one)")


@needs_two_gpus
def test_physical_gpus_rccl_group_materials(materials_scene):
    """gsp_multi_create_ex(GSP_GATHER_RCCL) over devices 0..n-1: ncclCommInitAll over DISTINCT devices, every share renders on
    its own GPU, one grouped ncclSend / ncclRecv brings the tiles to device 0 over xGMI; the frame equals one context's."""
    import gpuspectral_amd as g
    from gpuspectral_amd import abi, pt

    n = min(_gpus(), 8)
    W, H, spp = 416, 300, 8
    with g.Context(0) as ctx:
        ctx.upload_scene(materials_scene)
        ctx.frame_begin(W, H)
        ctx.render(spp=spp)
        ref = ctx.download()
        st1 = ctx.stats()
    with pt.MultiContext(list(range(n)), options=abi.CtxOptions(gather_route=abi.GATHER_RCCL)) as m:
        assert m.gather_route()[0] == "rccl"
        m.upload_scene(materials_scene)
        m.frame_begin(W, H)
        m.render(spp=spp)
        img = m.download()
        route, n_rccl, n_copy = m.gather_route()
        tot, each = m.stats(per_share=True)
        assert route == "rccl" and n_rccl >= 1 and n_copy == 0
        assert np.array_equal(img, ref)
        assert tot["extension_rays"] == st1["extension_rays"] and tot["shadow_rays"] == st1["shadow_rays"]
        assert len(each) == n and all(e["samples"] > 0 for e in each)
        m.frame_begin(W, H)  # a second frame through the same communicator
        m.render(spp=spp)
        assert np.array_equal(m.download(), ref)
    # GSP_GATHER_AUTO over distinct devices takes the RCCL route as well (the copy route is the fallback, not the default)
    with pt.MultiContext(list(range(n))) as m:
        assert m.gather_route()[0] == "rccl", m._L.gsp_multi_last_error(m._h).decode()


@needs_two_gpus
def test_physical_gpus_config4_frame():
    """BASELINE config 4 (bathroom2 stand-in, 1920x1080, image tiles over the GPUs of the node, RCCL gather) at 8 spp on
    min(devices, 8) physical GPUs: CRC of the gathered frame == the single-GPU frame's."""
    import gpuspectral_amd as g
    from gpuspectral_amd import abi, pt, scenes

    n = min(_gpus(), 8)
    sc = scenes.interior(600_000, seed=7)
    W, H, spp = 1920, 1080, 8
    with g.Context(0) as ctx:
        ctx.upload_scene(sc)
        ctx.frame_begin(W, H)
        ctx.render(spp=spp)
        ref = ctx.download()
        st1 = ctx.stats()
    with pt.MultiContext(list(range(n)), options=abi.CtxOptions(gather_route=abi.GATHER_RCCL)) as m:
        m.upload_scene(sc)
        m.frame_begin(W, H)
        m.render(spp=spp)
        img = m.download()
        tot = m.stats()
        assert m.gather_route()[0] == "rccl"
    assert zlib.crc32(img.tobytes()) == zlib.crc32(ref.tobytes())
    assert tot["extension_rays"] == st1["extension_rays"] and tot["shadow_rays"] == st1["shadow_rays"]


@needs_two_gpus
def test_physical_gpus_bench_over_nccl(tmp_path):
    """`bench.py --gpus n --backend nccl` (torch.distributed.run, one rank per GPU, RCCL over xGMI) for n = min(devices, 8):
    the dumped frame equals `--gpus 1`'s, every rank reports its share, and the line carries the gather's own time."""
    n = min(_gpus(), 8)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--steps", "1", "--warmup", "1", "--spp-per-step", "4", "--tris", "60000", "--width", "832", "--height", "480",
              "--no-cpu-baseline"]
    f1, f2 = str(tmp_path / "one.npy"), str(tmp_path / "many.npy")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dump", f1] + common, env=env,
                        capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-3000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                         "--master-port", "29677", os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--backend", "nccl", "--dump", f2] + common,
                        env=env, capture_output=True, text=True, timeout=1200)
    assert r2.returncode == 0, r2.stderr[-3000:]
    j1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    j2 = json.loads([l for l in r2.stdout.splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == n and j2["config"]["collective"].startswith("nccl, %d rank" % n)
    pr = j2["config"]["per_rank"]
    assert [p["rank"] for p in pr] == list(range(n)) and sum(p["pixels"] for p in pr) == 832 * 480 and all(p["samples"] > 0 for p in pr)
    assert j2["config"]["gather_ms"] is not None and j2["config"]["gather_ms"] > 0.0
    assert j1["config"]["path_segments"] == j2["config"]["path_segments"] and j1["config"]["shadow_rays"] == j2["config"]["shadow_rays"]
    a, b = np.load(f1), np.load(f2)
    assert a.shape == b.shape == (480, 832, 4) and np.array_equal(a, b)
