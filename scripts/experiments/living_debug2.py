import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import gpuspectral_amd as g
from gpuspectral_amd import abi
from oracle import oracle as orc
sc = abi.SceneArrays.load("tests/golden/ref_scenes/living-room.npz")
o = orc.Oracle(sc)
W, H = 320, 180
rays = o.extension_rays_of(W, H, spp=2)
want = o.trace(rays)
print("rays", len(rays), "oracle hits", int((want["prim"] >= 0).sum()))
with g.Context(0) as ctx:
    ctx.upload_scene(sc)
    for rep in range(4):
        got = ctx.trace(rays)
        bad = np.nonzero((got["prim"] != want["prim"]) | ((got["t"] != want["t"]) & (want["prim"] >= 0)))[0]
        print("rep", rep, "differing rays:", len(bad))
        for i in bad[:12]:
            print("  ray", i, rays[i].tolist(), "gpu", got[i], "oracle", want[i])
        # the same rays in another order
        perm = np.random.RandomState(rep).permutation(len(rays))
        got2 = ctx.trace(rays[perm])
        print("   permuted: differing from the first GPU answer:", int((got2["prim"] != got["prim"][perm]).sum()))
