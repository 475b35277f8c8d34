"""Cold against warm: the first render call of a context (pool allocation, first touches, lazy module loads) and the same call again."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gpuspectral_amd as g
from gpuspectral_amd import scenes
sc = scenes.interior(int(os.environ.get("TRIS", "1000000")), seed=int(os.environ.get("SEED", "7")))
for trial in range(2):
    with g.Context(0) as ctx:
        t = time.time(); ctx.upload_scene(sc); ctx.sync(); up = time.time() - t
        t = time.time(); ctx.frame_begin(1920, 1080); fb = time.time() - t
        out = []
        for rep in range(4):
            ctx.reset_stats(); t = time.time(); ctx.render(spp=16, first_timestamp=16 * rep); t1 = time.time() - t; ctx.sync(); dt = time.time() - t
            st = ctx.stats()
            out.append("%.0f ms (call returned after %.0f; %.0f Mrays/s)" % (dt * 1e3, t1 * 1e3, st["traced_rays"] / dt / 1e6))
        print("context %d: upload %.0f ms, frame_begin %.0f ms, 16-spp renders: %s" % (trial, up * 1e3, fb * 1e3, " | ".join(out)), flush=True)
