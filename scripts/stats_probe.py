"""Traversal statistics of the loaded build (GSP_LIB_PATH): nodes / triangles per extension and shadow ray."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpuspectral_amd as g
from gpuspectral_amd import scenes
sc = scenes.interior(1_000_000)
with g.Context(0) as ctx:
    ctx.upload_scene(sc); ctx.frame_begin(1920, 1080); ctx.reset_stats()
    ctx.render(spp=2, collect_traversal_stats=1)
    st = ctx.stats()
    print("ext: %.2f nodes/ray %.2f tris/ray (%d rays) | shadow: %.2f nodes/ray %.2f tris/ray (%d rays)" % (
        st["nodes_visited"] / max(1, st["stat_rays"]), st["tris_tested"] / max(1, st["stat_rays"]), st["stat_rays"],
        st["shadow_nodes_visited"] / max(1, st["shadow_stat_rays"]), st["shadow_tris_tested"] / max(1, st["shadow_stat_rays"]), st["shadow_stat_rays"]))
