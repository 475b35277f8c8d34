import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import gpuspectral_amd as g
from gpuspectral_amd import abi
from oracle import oracle as orc
from conftest import random_rays
sc = abi.SceneArrays.load("tests/golden/ref_scenes/living-room.npz")
o = orc.Oracle(sc)
W, H = 320, 180
ref, ost = o.render(W, H, spp=2)
with g.Context(0) as ctx:
    ctx.upload_scene(sc)
    if "trace" in sys.argv:
        lo, hi = sc.positions.min(0), sc.positions.max(0)
        rays = random_rays(300000, 23, lo=tuple(lo - 0.1), hi=tuple(hi + 0.1))
        ctx.trace(rays)
    for rep in range(3):
        ctx.frame_begin(W, H)
        ctx.reset_stats()
        ctx.render(spp=2, collect_kernel_times=1)
        img = ctx.download().reshape(-1, 4)
        st = ctx.stats()
        print("rep", rep, "ext", st["extension_rays"], "oracle", ost["extension_rays"], "shaded", st["shaded_vertices"], ost.get("shaded_vertices"), "memoised", st.get("memoised_rays"), "img equal", np.array_equal(img, ref), "depth/nodes", st["num_bvh_nodes"], flush=True)
