"""Throughput and build time against scene size: the bench workload (scenes.interior, 1920x1080) at 0.25 / 1 / 4 / 16 M triangles.
python scripts/size_scaling_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpuspectral_amd as g
from gpuspectral_amd import scenes

print("# %-10s %10s %9s %6s %10s %12s %10s %10s %10s" % ("triangles", "nodes", "build ms", "depth", "Mrays/s", "Msamples/s", "extend ms", "shade ms", "connect ms"))
for tris in (250_000, 1_000_000, 4_000_000, 16_000_000):
    sc = scenes.interior(tris, seed=7)
    with g.Context(0) as ctx:
        ctx.upload_scene(sc)
        t = time.time(); ctx.upload_scene(sc); build = 1e3 * (time.time() - t)  # (second build of the process: no one-time costs)
        ctx.frame_begin(1920, 1080)
        ctx.render(spp=16); ts = 16
        ctx.sync()
        ctx.reset_stats(); t = time.time(); ctx.render(spp=48, first_timestamp=ts, collect_kernel_times=1); ctx.sync(); dt = time.time() - t
        st = ctx.stats()
        print("  %-10d %10d %9.1f %6d %10.0f %12.1f %10.1f %10.1f %10.1f" % (sc.num_triangles, st["num_bvh_nodes"], build, st["bvh_depth"], st["traced_rays"] / dt / 1e6,
                                                                         st["samples"] / dt / 1e6, st["extend_kernel_ms"], st["shade_kernel_ms"], st["connect_kernel_ms"]), flush=True)
