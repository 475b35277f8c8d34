"""Hand-run: what the dormant-feature extension costs on the bench scene (1M triangles, 1080p): the same scene plain,
with every texturable BSDF textured (k_shade<true>, per-vertex uv + four texel fetches), and textured + sky."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np  # noqa: E402

import gpuspectral_amd as g  # noqa: E402
import textured  # noqa: E402
from gpuspectral_amd import abi, scenes  # noqa: E402


def run(sc, label):
    with g.Context(0) as ctx:
        ctx.upload_scene(sc)
        ctx.frame_begin(1920, 1080)
        ctx.render(spp=8)
        ts, best = 8, None
        for _ in range(3):
            ctx.reset_stats()
            t = time.time()
            ctx.render(spp=48, first_timestamp=ts, collect_kernel_times=1)
            ctx.sync()
            dt = time.time() - t
            ts += 48
            st = ctx.stats()
            r = (st["traced_rays"] / dt / 1e6, st["extend_kernel_ms"], st["shade_kernel_ms"], st["connect_kernel_ms"])
            best = r if best is None or r[0] > best[0] else best
        print("%-34s %.1f Mrays/s | extend %.1f shade %.1f connect %.1f ms | device %.2f GB" % ((label,) + best + (st["device_bytes"] / 1e9,)), flush=True)


for rep in range(2):
    run(scenes.interior(1_000_000), "plain (k_shade<false>)")
    sc = scenes.interior(1_000_000)
    rng = np.random.RandomState(1)
    sc.uvs = rng.uniform(0, 4, (len(sc.positions), 2)).astype(np.float32)
    for size in (2048, 1024, 512):
        sc.add_texture(rng.randint(60, 256, (size, size, 4)).astype(np.uint8))
    for name in ("diffuse", "rough_conductor", "rough_plastic"):
        recs = sc.bsdfs[abi.BSDF_NAMES.index(name)]
        recs["has_texture"] = 1 + (np.arange(len(recs)) % 3)
    sc.texel_decode = textured.srgb_table()
    run(sc, "all texturable BSDFs textured")
    sc.env_texels = rng.uniform(0, 1, (512, 1024, 4)).astype(np.float32)
    run(sc, "textured + environment map")
