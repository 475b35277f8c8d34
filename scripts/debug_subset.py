import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gpuspectral_amd as g
from gpuspectral_amd import scenes, multigpu
sc = scenes.interior(1_000_000)
W, H = 1920, 1080
with g.Context(0) as ctx:
    ctx.upload_scene(sc)
    for spp, K in ((1, 0), (8, 0), (17, 0)):
        ctx.frame_begin(W, H); ctx.render(spp=spp, timestamps_in_flight=K); full = ctx.download().reshape(-1, 4).copy()
        ctx.frame_begin(W, H); ctx.render(spp=spp, timestamps_in_flight=K); full2 = ctx.download().reshape(-1, 4).copy()
        print("spp", spp, "full deterministic:", np.array_equal(full, full2))
        out = np.zeros_like(full)
        for r in range(2):
            ids = multigpu.partition(W, H, r, 2)
            ctx.frame_begin(W, H, ids); ctx.render(spp=spp, timestamps_in_flight=K); out[ids] = ctx.download_compact()
        d = np.abs(out - full).max(1)
        print("   subset==full:", np.array_equal(out, full), "ndiff", (d > 0).sum(), "max", d.max())
        if (d > 0).any():
            bad = np.argwhere(d > 0).ravel()[:8]; print("   bad px", [(int(b % W), int(b // W)) for b in bad])
