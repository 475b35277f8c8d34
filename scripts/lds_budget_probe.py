"""What would an LDS copy of BVH nodes / triangle packets capture?    For each scene: visits per node
record and tests per triangle slot of the closest-hit rays of a few samples (collect_traversal_stats = 2), then the share
of all visits / tests that goes to the first K records of the level-ordered node array (what k_trace stages: kTopNodes) and
to the K most-visited records (the best any static choice of K could do).   python scripts/lds_budget_probe.py [scenes]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpuspectral_amd as g
from gpuspectral_amd import abi, scenes


def load(name):
    if name == "interior":
        return scenes.interior(1_000_000, seed=7)
    if name == "materials":
        return scenes.cornell_materials(96)
    if name == "caustics":
        return scenes.caustics(1_000_000, seed=11)
    return abi.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "ref_scenes", name + ".npz"))


with g.Context(0) as ctx:
    for name in sys.argv[1:] or ["interior", "coffee", "staircase2", "materials", "caustics"]:
        ctx.upload_scene(load(name))
        ctx.frame_begin(960, 540)
        ctx.reset_stats()
        ctx.render(spp=4, collect_traversal_stats=2)
        st = ctx.stats()
        nodes, slots = ctx.visit_histograms()
        rays = max(1, st["stat_rays"])
        out = dict(scene=name, triangles=st["num_triangles"], nodes=st["num_bvh_nodes"], rays=rays,
                   nodes_per_ray=round(nodes.sum() / rays, 2), tris_per_ray=round(slots.sum() / rays, 2))
        ns, ts = np.sort(nodes)[::-1].astype(np.float64), np.sort(slots)[::-1].astype(np.float64)
        for k in (16, 64, 128, 256, 512):
            out["first_%d_nodes" % k] = round(float(nodes[:k].sum()) / rays, 2)   # visits per ray an LDS copy of the first k captures
            out["best_%d_nodes" % k] = round(float(ns[:k].sum()) / rays, 2)
        first = 4  # (kFirstSlot: the leading all-zero slots)
        for k in (16, 64, 256, 1024):
            out["best_%d_tris" % k] = round(float(ts[:k].sum()) / rays, 3)       # tests per ray the k most-tested packets capture
            out["first_%d_tris" % k] = round(float(slots[first:first + k].sum()) / rays, 3)  # ... and the first k slots (level order)
        print(json.dumps(out), flush=True)
