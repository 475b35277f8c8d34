"""Per-scene throughput + traversal statistics on the GPU (no oracle): Mrays/s over >= ~4 s of work, kernel time split,
BVH nodes / triangles per ray.  Scenes: the reference's shipped ones (tests/golden/ref_scenes/*.npz) and the generated
stand-ins.   python tests/tools/scene_probe.py [coffee staircase2 cornell-box interior materials caustics living-room_shapes staircase2_shapes]
(`*_shapes`: the same scene loaded with LoadOptions::builtinShapes -- its `disk` / `sphere` emitters built instead of skipped)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpuspectral_amd as g
from gpuspectral_amd import abi, scenes

W, H = int(os.environ.get("W", "1280")), int(os.environ.get("H", "720"))
SECONDS = float(os.environ.get("SECONDS", "4"))


def load(name):
    if name == "interior":
        return scenes.interior(1_000_000, seed=7)
    if name == "materials":
        return scenes.cornell_materials(96)
    if name == "caustics":
        return scenes.caustics(1_000_000, seed=11)
    return abi.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "ref_scenes", name + ".npz"))


with g.Context(0) as ctx:
    for name in sys.argv[1:] or ["coffee", "staircase2", "cornell-box", "interior"]:
        sc = load(name)
        ctx.upload_scene(sc)
        ctx.frame_begin(W, H)
        ctx.render(spp=16); ctx.sync(); ts = 16
        ctx.reset_stats(); ctx.render(spp=2, first_timestamp=ts, collect_traversal_stats=1); ts += 2
        s0 = ctx.stats()
        # size the run: probe 32 spp, then ~SECONDS
        ctx.reset_stats(); t = time.time(); ctx.render(spp=32, first_timestamp=ts); ctx.sync(); dt = time.time() - t; ts += 32
        spp = int(max(64, min(16384, 32 * SECONDS / max(dt, 1e-3))))
        ctx.reset_stats(); t = time.time(); ctx.render(spp=spp, first_timestamp=ts, collect_kernel_times=1); ctx.sync(); dt = time.time() - t
        st = ctx.stats()
        rays = st["traced_rays"]
        print(json.dumps(dict(scene=name, triangles=st["num_triangles"], bvh_nodes=st["num_bvh_nodes"], resolution="%dx%d" % (W, H), spp=spp,
                              seconds=round(dt, 3), mrays_per_s=round(rays / dt / 1e6, 1), msamples_per_s=round(st["samples"] / dt / 1e6, 1),
                              rays_per_sample=round(rays / st["samples"], 2), shadow_share=round(st["shadow_rays"] / rays, 3),
                              extend_ms=round(st["extend_kernel_ms"], 1), shade_ms=round(st["shade_kernel_ms"], 1), connect_ms=round(st["connect_kernel_ms"], 1),
                              ns_per_ext_ray=round(st["extend_kernel_ms"] * 1e6 / st["extension_rays"], 4),
                              ns_per_shadow_ray=round(st["connect_kernel_ms"] * 1e6 / max(1, st["shadow_rays"]), 4),
                              ns_per_vertex=round(st["shade_kernel_ms"] * 1e6 / max(1, st["shaded_vertices"]), 4),
                              launches=st["extend_launches"],
                              nodes_per_ray=round(s0["nodes_visited"] / max(1, s0["stat_rays"]), 2), tris_per_ray=round(s0["tris_tested"] / max(1, s0["stat_rays"]), 2),
                              shadow_nodes_per_ray=round(s0["shadow_nodes_visited"] / max(1, s0["shadow_stat_rays"]), 2),
                              shadow_tris_per_ray=round(s0["shadow_tris_tested"] / max(1, s0["shadow_stat_rays"]), 2),
                              # any-hit rays by verdict (is an occluder cache worth building?)
                              shadow_occluded_share=round(s0["shadow_stat_occluded"] / max(1, s0["shadow_stat_rays"]), 3),
                              # shadow rays that never reached a triangle test (would a lazy shear set-up pay?)
                              shadow_no_triangle_share=round(s0["shadow_stat_no_triangle"] / max(1, s0["shadow_stat_rays"]), 3),
                              occluded_nodes_per_ray=round(s0["shadow_stat_occluded_nodes"] / max(1, s0["shadow_stat_occluded"]), 2),
                              unoccluded_nodes_per_ray=round((s0["shadow_nodes_visited"] - s0["shadow_stat_occluded_nodes"]) /
                                                             max(1, s0["shadow_stat_rays"] - s0["shadow_stat_occluded"]), 2))), flush=True)
