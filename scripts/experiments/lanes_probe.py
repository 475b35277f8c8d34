"""Wall-clock throughput of SPP-sample calls on the bench scene (no per-kernel times: with GSP_LANES=2 kernels of two streams overlap)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gpuspectral_amd as g
from gpuspectral_amd import scenes
sc = scenes.interior(1_000_000)
SPP = int(os.environ.get("SPP", "256"))
with g.Context(0) as ctx:
    ctx.upload_scene(sc); ctx.frame_begin(1920, 1080); ctx.render(spp=8); ts = 8
    out = []
    for rep in range(int(os.environ.get("REPS", "3"))):
        ctx.reset_stats(); t = time.time(); ctx.render(spp=SPP, first_timestamp=ts); ctx.sync(); dt = time.time() - t; ts += SPP
        st = ctx.stats()
        out.append("%.0f" % (st["traced_rays"] / dt / 1e6))
    print("Mrays/s per %d-spp call: %s | Msamples/s %.1f" % (SPP, " ".join(out), st["samples"] / dt / 1e6), flush=True)
