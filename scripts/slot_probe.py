import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpuspectral_amd as g
from gpuspectral_amd import scenes
sc = scenes.interior(1_000_000)
for kb in (1, 2, 4, 8, 1):
    with g.Context(0) as ctx:
        ctx.upload_scene(sc); ctx.frame_begin(1920, 1080); ctx.render(spp=16, timestamps_in_flight=kb); ts = 16
        best = None
        for rep in range(2):
            ctx.reset_stats(); t = time.time(); ctx.render(spp=48, first_timestamp=ts, collect_kernel_times=1, timestamps_in_flight=kb); ctx.sync(); dt = time.time() - t; ts += 48
            st = ctx.stats()
            r = (st["traced_rays"] / dt / 1e6, st["extend_kernel_ms"], st["shade_kernel_ms"], st["connect_kernel_ms"], st["extend_launches"])
            best = r if best is None or r[0] > best[0] else best
        print("timestamps per batch %d: %.1f Mrays/s | extend %.1f shade %.1f connect %.1f ms, %d launches" % ((kb,) + best), flush=True)
