"""Robustness at the edges of the BASELINE configs: a 10 M-triangle scene and a full 4096x4096 frame on one GPU,
each with a sparse oracle parity check."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpuspectral_amd as g
from gpuspectral_amd import scenes, abi
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O

def run(name, sc, W, H, spp, check_px=64, **kw):
    with g.Context(0) as ctx:
        t = time.time(); ctx.upload_scene(sc); up = time.time() - t
        ctx.frame_begin(W, H); ctx.reset_stats()
        t = time.time(); ctx.render(spp=spp, **kw); ctx.sync(); dt = time.time() - t
        st = ctx.stats()
        img = ctx.download().reshape(-1, 4)
        pick = np.linspace(0, W * H - 1, check_px).astype(np.uint32)
        p = abi.default_render_params(spp, 0)
        for k, v in kw.items():
            setattr(p, k, v)
        ref, _ = O.Oracle(sc).render(W, H, spp=spp, pixel_ids=pick, params=p)
        print(json.dumps(dict(case=name, triangles=st["num_triangles"], resolution="%dx%d" % (W, H), spp=spp, seconds=dt,
                              mrays_per_s=st["traced_rays"] / dt / 1e6, upload_build_ms=up * 1e3,
                              device_gb=st["device_bytes"] / 1e9, oracle_pixels_checked=check_px,
                              oracle_pixels_differing=int((img[pick] != ref).any(1).sum()), nan_pixels=int(np.isnan(img).any(1).sum()))), flush=True)

run("10M-triangle interior, 1080p", scenes.interior(10_000_000, seed=3), 1920, 1080, 16)
run("full 4096x4096 caustics frame, max_depth 32", scenes.caustics(1_000_000), 4096, 4096, 8, max_depth=32)
