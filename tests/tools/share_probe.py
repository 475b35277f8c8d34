"""What one GPU of an 8-GPU job sees: Mrays/s on the 1/8 tile share of the 1080p bench frame against the full frame
(same scene, same total samples), plus the GPU-busy share of the wall time.  SURVEY 8(e) readiness without 8 GPUs."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpuspectral_amd as g
from gpuspectral_amd import scenes, multigpu

W, H = 1920, 1080
sc = scenes.interior(1_000_000, seed=7)
with g.Context(0) as ctx:
    ctx.upload_scene(sc)
    for world in (1, 2, 4, 8):
        ids = multigpu.partition(W, H, 0, world)
        ctx.frame_begin(W, H, ids)
        spp = 64 * world
        ctx.render(spp=spp); ctx.sync()
        best = None
        for rep in range(2):
            ctx.reset_stats(); t = time.time(); ctx.render(spp=4 * spp, first_timestamp=spp * (1 + 4 * rep), collect_kernel_times=1); ctx.sync(); dt = time.time() - t
            st = ctx.stats()
            r = dict(share="1/%d" % world, pixels=ctx.num_pixels, spp=4 * spp, seconds=round(dt, 3),
                     mrays_per_s=round(st["traced_rays"] / dt / 1e6, 1),
                     kernels_ms=round(st["extend_kernel_ms"] + st["shade_kernel_ms"] + st["connect_kernel_ms"], 1),
                     busy=round((st["extend_kernel_ms"] + st["shade_kernel_ms"] + st["connect_kernel_ms"]) / (dt * 1e3), 4),
                     launches=st["extend_launches"], rays_per_launch=round(st["extension_rays"] / max(1, st["extend_launches"]) / 1e6, 2))
            best = r if best is None or r["mrays_per_s"] > best["mrays_per_s"] else best
        print(json.dumps(best), flush=True)
