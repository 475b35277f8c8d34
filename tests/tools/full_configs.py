"""BASELINE.json configs 2, 3 and 5 at their FULL resolution and sample counts on one GPU (config 5: the 1/8 tile
share one rank of the 8-GPU job owns).  For each: throughput, and -- at the full spp -- bit-parity of a sparse pixel
subset against the oracle (the seed is a function of (pixel, timestamp) only, so a subset reproduces the frame)."""
import os, sys, time, json, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpuspectral_amd as g
from gpuspectral_amd import scenes, abi
from gpuspectral_amd.scenes import tile_pixel_ids
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O

def run(name, sc, W, H, spp, ids=None, check_px=96, **kw):
    with g.Context(0) as ctx:
        t = time.time(); ctx.upload_scene(sc); up = time.time() - t
        ctx.frame_begin(W, H, ids); ctx.render(spp=2, **kw)
        ctx.frame_begin(W, H, ids); ctx.reset_stats()
        t = time.time(); ctx.render(spp=spp, **kw); ctx.sync(); dt = time.time() - t
        st = ctx.stats()
        img = ctx.download_compact()
        own = np.arange(W * H, dtype=np.uint32) if ids is None else ids
        pick = np.linspace(0, len(own) - 1, check_px).astype(np.int64)
        p = abi.default_render_params(spp, 0)
        for k, v in kw.items():
            setattr(p, k, v)
        t = time.time(); ref, _ = O.Oracle(sc).render(W, H, spp=spp, pixel_ids=own[pick], params=p); cpu_dt = time.time() - t
        ndiff = int((img[pick] != ref).any(1).sum())
        rec = dict(config=name, triangles=st["num_triangles"], resolution="%dx%d" % (W, H), pixels=int(len(own)), spp=spp, seconds=dt,
                   mrays_per_s=st["traced_rays"] / dt / 1e6, msamples_per_s=st["samples"] / dt / 1e6,
                   rays_per_sample=st["traced_rays"] / st["samples"], upload_build_ms=up * 1e3,
                   device_gb=st["device_bytes"] / 1e9, oracle_pixels_checked_at_full_spp=check_px, oracle_pixels_differing=ndiff,
                   oracle_seconds=cpu_dt, nan_pixels=int(np.isnan(img).any(1).sum()), frame_crc32="%08x" % zlib.crc32(img.tobytes()),
                   mean_rgb=[float(v) for v in img[:, :3].mean(0)], **{k: v for k, v in kw.items()})
        print(json.dumps(rec), flush=True)

which = sys.argv[1:] or ["2", "3a", "3b", "5"]
if "2" in which:
    run("config 2: cornell_materials(96), full BSDF set", scenes.cornell_materials(96), 1024, 1024, 1024)
if "3a" in which:
    run("config 3: interior(600k)", scenes.interior(600_000), 1920, 1080, 4096)
if "3b" in which:
    run("config 3 headline: interior(1M, seed 7)", scenes.interior(1_000_000, seed=7), 1920, 1080, 4096)
if "5" in which:
    run("config 5 (rank 0 of 8): caustics(1M), max_depth 32", scenes.caustics(1_000_000), 4096, 4096, 8192,
        ids=tile_pixel_ids(4096, 4096, 0, 8), check_px=48, max_depth=32)
