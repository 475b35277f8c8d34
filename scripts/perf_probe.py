"""Quick GPU throughput probe (not the bench contract): prints stats for a few scenes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np

import gpuspectral_amd as g
from gpuspectral_amd import scenes

def probe(name, sc, W, H, spp, K=0):
    with g.Context(0) as ctx:
        t = time.time(); ctx.upload_scene(sc); up = time.time() - t
        ctx.frame_begin(W, H)
        ctx.render(spp=min(spp, 2))  # warmup
        ctx.reset_stats()
        t = time.time()
        ctx.render(spp=spp, first_timestamp=2, timestamps_in_flight=K, collect_kernel_times=1)
        ctx.sync()
        dt = time.time() - t
        st = ctx.stats()
        rays = st["traced_rays"]
        print("%s: %d tris, upload+build %.1f ms (build %.1f), %dx%d x %d spp: %.3f s, %.1f Mrays/s, %.2f Msamples/s | "
              "extend %.1f ms (%d launches) shade %.1f ms connect %.1f ms | ext %d sh %d | mem %.2f GB"
              % (name, st["num_triangles"], up * 1e3, st["bvh_build_ms"], W, H, spp, dt, rays / dt / 1e6,
                 st["samples"] / dt / 1e6, st["extend_kernel_ms"], st["extend_launches"], st["shade_kernel_ms"],
                 st["connect_kernel_ms"], st["extension_rays"], st["shadow_rays"], st["device_bytes"] / 1e9), flush=True)
        ctx.reset_stats()
        ctx.render(spp=1, first_timestamp=100, collect_traversal_stats=1)
        st = ctx.stats()
        print("   traversal: %.1f nodes/ray, %.2f tris/ray over %d rays | shadow: %.1f nodes/ray, %.2f tris/ray over %d rays" % (
            st["nodes_visited"] / max(1, st["stat_rays"]), st["tris_tested"] / max(1, st["stat_rays"]), st["stat_rays"],
            st["shadow_nodes_visited"] / max(1, st["shadow_stat_rays"]), st["shadow_tris_tested"] / max(1, st["shadow_stat_rays"]), st["shadow_stat_rays"]), flush=True)

if __name__ == "__main__":
    which = sys.argv[1:] or ["cornell", "mats", "interior"]
    if "cornell" in which:
        probe("cornell_materials(16)", scenes.cornell_materials(16), 1024, 1024, 16)
    if "mats" in which:
        probe("cornell_materials(64)", scenes.cornell_materials(64), 1024, 1024, 16)
    if "interior" in which:
        probe("interior(600k)", scenes.interior(600_000), 1920, 1080, 8)
    if "interior1m" in which:
        probe("interior(1M)", scenes.interior(1_000_000), 1920, 1080, 8)
    if "caustics" in which:
        probe("caustics(200k)", scenes.caustics(200_000), 1024, 1024, 8)
