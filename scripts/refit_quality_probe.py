"""What a refitted tree costs the traversal: the 988 k-triangle bench scene at 1280x720, an object moved by d, rendered once
through the tree gsp_update_instances refits (gsp_ctx_options.refit_growth = 1e9: always refit) and once through a rebuilt
one (refit_growth = 1: always rebuild).  Prints Mrays/s of both, the update's duration, and what the default bound (1.25)
decides.   python scripts/refit_quality_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpuspectral_amd as g
from gpuspectral_amd import abi, scenes

W, H, SPP = 1280, 720, 96


def rate(ctx):
    ctx.frame_begin(W, H)
    ctx.render(spp=8)
    ctx.sync()
    ctx.reset_stats()
    t = time.time()
    ctx.render(spp=SPP, first_timestamp=8)
    ctx.sync()
    dt = time.time() - t
    st = ctx.stats()
    return (st["extension_rays"] - st["memoised_rays"] + st["memo_build_rays"] + st["shadow_rays"]) / dt / 1e6


sc = scenes.interior(1_000_000, seed=7)
base = sc.instances.copy()
order = np.argsort(-base["vertex_count"].astype(np.int64))
print("# %d triangles, %d instances; moved: the k largest objects, each by d along (1, 0.3, -0.5) (scene units; the room is ~10 across)" % (sc.num_triangles, len(base)))
print("# %-28s %12s %12s %10s %10s   default bound" % ("edit", "refit Mrays/s", "rebuild", "refit ms", "rebuild ms"))
ctxs = {k: g.Context(0, options=abi.CtxOptions(refit_growth=v, memory_share=0.25)) for k, v in (("refit", 1e9), ("rebuild", 1.0), ("default", 0.0))}
try:
    for c in ctxs.values():
        c.upload_scene(sc)
    for name in ("refit", "rebuild"):
        rate(ctxs[name])  # (first render of a context: pool allocation)
    print("  %-28s %12.0f %12.0f" % ("as built", rate(ctxs["refit"]), rate(ctxs["rebuild"])))
    for k, d in ((1, 0.05), (1, 0.5), (1, 2.0), (8, 0.05), (8, 0.5), (8, 2.0), (len(base), 0.5)):
        inst = base.copy()
        for j in order[:k]:
            t = inst["transform"][j].copy()
            t[12:15] += np.float32(d) * np.array([1.0, 0.3, -0.5], np.float32) * (1.0 if j % 2 else -1.0)
            inst["transform"][j] = t
        out = {}
        for name in ("refit", "rebuild", "default"):
            c = ctxs[name]
            c.update_instances(base)  # back to the built state (a refit / rebuild of its own)
            if name != "rebuild":  # measure the edit against a freshly BUILT tree
                c.upload_scene(sc)
            r0 = c.stats()["scene_refits"]
            t0 = time.time()
            c.update_instances(inst)
            ms = 1e3 * (time.time() - t0)
            out[name] = (rate(c) if name != "default" else 0.0, ms, c.stats()["scene_refits"] - r0)
        print("  %-28s %12.0f %12.0f %10.2f %10.2f   %s" % ("%d object(s) by %.2f" % (k, d), out["refit"][0], out["rebuild"][0], out["refit"][1], out["rebuild"][1],
                                                    "refits" if out["default"][2] else "rebuilds"))
finally:
    for c in ctxs.values():
        c.close()
