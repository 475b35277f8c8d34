"""Traversal statistics + throughput of the bench scene for GSP_BVH_REINSERT=rounds (the env of the caller)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gpuspectral_amd as g
from gpuspectral_amd import scenes
name = sys.argv[1] if len(sys.argv) > 1 else "interior"
sc = scenes.interior(1_000_000) if name == "interior" else scenes.caustics(1_000_000, seed=11)
with g.Context(0) as ctx:
    t = time.time(); ctx.upload_scene(sc); ctx.sync(); up = time.time() - t
    t = time.time(); ctx.upload_scene(sc); ctx.sync(); up2 = time.time() - t
    ctx.frame_begin(1920, 1080); ctx.reset_stats()
    ctx.render(spp=2, collect_traversal_stats=1)
    st = ctx.stats()
    line = "ext %.2f nodes %.2f tris | shadow %.2f / %.2f | %d nodes depth %d | upload %.0f / %.0f ms (build %.1f)" % (
        st["nodes_visited"] / max(1, st["stat_rays"]), st["tris_tested"] / max(1, st["stat_rays"]),
        st["shadow_nodes_visited"] / max(1, st["shadow_stat_rays"]), st["shadow_tris_tested"] / max(1, st["shadow_stat_rays"]),
        st["num_bvh_nodes"], st["bvh_depth"], up * 1e3, up2 * 1e3, st.get("bvh_build_ms", -1))
    ctx.render(spp=8, first_timestamp=2); ts = 10
    best = None
    for rep in range(2):
        ctx.reset_stats(); t = time.time(); ctx.render(spp=48, first_timestamp=ts, collect_kernel_times=1); ctx.sync(); dt = time.time() - t; ts += 48
        st = ctx.stats()
        r = (st["traced_rays"] / dt / 1e6, st["extend_kernel_ms"], st["shade_kernel_ms"], st["connect_kernel_ms"])
        best = r if best is None or r[0] > best[0] else best
    print(line + " | %.1f Mrays/s | extend %.1f shade %.1f connect %.1f ms" % best, flush=True)
