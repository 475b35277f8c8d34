import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpuspectral_amd as g
from gpuspectral_amd import scenes
sc = scenes.interior(1_000_000, seed=7)
with g.Context(0) as ctx:
    for k in range(5):
        t = time.time(); ctx.upload_scene(sc); dt = time.time() - t
        print("upload %d: bvh_build_ms %.2f, gsp_upload_scene %.2f ms" % (k, ctx.stats()["bvh_build_ms"], 1e3 * dt), flush=True)
    print("nodes %d, depth %d" % (ctx.stats()["num_bvh_nodes"], ctx.stats()["bvh_depth"]))
