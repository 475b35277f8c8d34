"""render the reference's shipped scenes (flattened by scripts/export_reference_scenes.py into
scene_cache/) on the GPU: rate, bit-parity against the oracle on a tile subset, and a tone-mapped PNG."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpuspectral_amd as g
from gpuspectral_amd import abi, host
from gpuspectral_amd.scenes import tile_pixel_ids
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O

W, H = 1280, 720
SPP = int(os.environ.get("SPP", "256"))
out_dir = os.path.join(ROOT, "gpurun_out", "ref_scenes")
os.makedirs(out_dir, exist_ok=True)
with g.Context(0) as ctx:
    for name in sys.argv[1:] or ["cornell-box", "coffee", "staircase2", "living-room"]:
        sc = abi.SceneArrays.load(os.path.join(ROOT, "scene_cache", name + ".npz"))
        t = time.time(); ctx.upload_scene(sc); up = time.time() - t
        # parity first: 8 spp on every 48th 32x32 tile, GPU subset context vs oracle
        ids = tile_pixel_ids(W, H, 0, 48)
        ctx.frame_begin(W, H, ids); ctx.reset_stats(); ctx.render(spp=8); sub = ctx.download_compact()
        ref, st_o = O.Oracle(sc).render(W, H, spp=8, pixel_ids=ids)
        st_g = ctx.stats()
        ndiff = int((sub != ref).any(1).sum())
        ctx.frame_begin(W, H); ctx.render(spp=4); ctx.reset_stats()
        t = time.time(); ctx.render(spp=SPP, first_timestamp=4); ctx.sync(); dt = time.time() - t
        st = ctx.stats()
        img = ctx.download()
        full_sub = img.reshape(-1, 4)[ids]
        rec = dict(scene=name, triangles=st["num_triangles"], lights=len(sc.lights), resolution="%dx%d" % (W, H), spp=SPP,
                   seconds=dt, mrays_per_s=st["traced_rays"] / dt / 1e6,
                   msamples_per_s=st["samples"] / dt / 1e6, rays_per_sample=st["traced_rays"] / st["samples"],
                   upload_build_ms=up * 1e3, parity_pixels=len(ids), parity_pixels_differing=ndiff,
                   parity_rays_equal=bool(st_g["extension_rays"] == st_o["extension_rays"] and st_g["shadow_rays"] == st_o["shadow_rays"]),
                   nan_pixels=int(np.isnan(img).any(2).sum()), mean_rgb=[float(v) for v in img[..., :3].mean((0, 1))])
        print(json.dumps(rec), flush=True)
        try:
            from PIL import Image
            Image.fromarray(host.tone_map(img, aces=False)).resize((640, 360)).save(os.path.join(out_dir, name + ".png"))
        except Exception as e:  # the image is a convenience, not a result
            print("png skipped:", e)
