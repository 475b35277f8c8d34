"""The 4096 x 4096 caustics frame of tests/tools/stress_probe.py, 8 spp per call, several calls: where does a call's time go?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gpuspectral_amd as g
from gpuspectral_amd import scenes
sc = scenes.caustics(1_000_000)
W = int(os.environ.get("W", "4096"))
with g.Context(0) as ctx:
    ctx.upload_scene(sc); ctx.sync()
    ctx.frame_begin(W, W)
    for rep in range(4):
        ctx.reset_stats(); t = time.time(); ctx.render(spp=8, first_timestamp=8 * rep, max_depth=32, collect_kernel_times=1); t1 = time.time() - t; ctx.sync(); dt = time.time() - t
        st = ctx.stats()
        print("call %d: %.0f ms (returned after %.0f) %.0f Mrays/s | kernels extend %.1f shade %.1f connect %.1f ms, %d launches | device %.1f GB" % (
            rep, dt * 1e3, t1 * 1e3, st["traced_rays"] / dt / 1e6, st["extend_kernel_ms"], st["shade_kernel_ms"], st["connect_kernel_ms"], st["extend_launches"], st["device_bytes"] / 1e9), flush=True)
