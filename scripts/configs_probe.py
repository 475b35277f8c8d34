"""BASELINE.json configs 2 and 5 at full resolution (reduced spp): throughput for DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpuspectral_amd as g
from gpuspectral_amd import scenes
def run(name, sc, W, H, spp, **kw):
    with g.Context(0) as ctx:
        ctx.upload_scene(sc); ctx.frame_begin(W, H); ctx.render(spp=4, **kw); ctx.reset_stats()
        t = time.time(); ctx.render(spp=spp, first_timestamp=4, **kw); ctx.sync(); dt = time.time() - t
        st = ctx.stats()
        print("%s: %d tris, %dx%d x %d spp: %.2f s, %.1f Mrays/s, %.1f Msamples/s, %.2f rays/sample, mem %.1f GB" % (
            name, st["num_triangles"], W, H, spp, dt, st["traced_rays"] / dt / 1e6,
            st["samples"] / dt / 1e6, st["traced_rays"] / st["samples"], st["device_bytes"] / 1e9), flush=True)
run("config 2: cornell_materials(96) full BSDF set", scenes.cornell_materials(96), 1024, 1024, 256)
run("config 3: interior(600k)", scenes.interior(600_000), 1920, 1080, 64)
run("config 5: caustics(1M) max_depth 32", scenes.caustics(1_000_000), 4096, 4096, 16, max_depth=32)
