import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpuspectral_amd as g
from gpuspectral_amd import scenes
sc = scenes.interior(1_000_000)
with g.Context(0) as ctx:
    ctx.upload_scene(sc); ctx.frame_begin(1920, 1080); ctx.render(spp=4); ts = 4
    for spp in (8, 16, 32, 64, 128):
        ctx.reset_stats(); t = time.time(); ctx.render(spp=spp, first_timestamp=ts); ctx.sync(); dt = time.time() - t; ts += spp
        st = ctx.stats()
        print("spp/call %3d: %.3f s  %.1f Mrays/s  %.1f Msamples/s" % (spp, dt, (st["extension_rays"] + st["shadow_rays"]) / dt / 1e6, st["samples"] / dt / 1e6), flush=True)
