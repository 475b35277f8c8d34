"""The BASELINE.json configs at their stated sizes, shared by the golden generator
(tests/golden/make_full_configs.py, oracle side) and the GPU tests (tests/test_gpu_full_configs.py, HIP side).

No oracle import here: the GPU tests read the committed expectations, they do not run the oracle at these sizes.
"""
import numpy as np

from gpuspectral_amd import abi, scenes

SPP_QUICK = 16   # short sample count checked in addition to the full one (and used for the determinism re-run)
SUBSET = 256     # pixels of each frame the oracle renders (BASELINE asks for per-pixel parity; >= 256 per config)

# name -> scene generator call, frame, full sample count, integrator overrides, tile share (rank, world) or None
CONFIGS = {
    # config 2: Cornell box + full BSDF set, 1024x1024, 1024 spp
    "config2": dict(scene=("cornell_materials", (96,), {}), width=1024, height=1024, spp=1024, params={}, share=None),
    # config 3: bathroom2 stand-in (~600k triangles), 1920x1080, 4096 spp
    "config3": dict(scene=("interior", (600_000,), {"seed": 7}), width=1920, height=1080, spp=4096, params={}, share=None),
    # headline variant of config 3: ~1M triangles (bench.py's workload)
    "headline": dict(scene=("interior", (1_000_000,), {"seed": 7}), width=1920, height=1080, spp=4096, params={}, share=None),
    # config 5: dielectric caustics, max bounce depth 32, 4096x4096, 8192 spp -- one GPU renders the tile share
    # rank 0 of the 8-GPU job owns (1/8 of the pixels, interleaved 32x32 tiles)
    "config5": dict(scene=("caustics", (1_000_000,), {"seed": 11}), width=4096, height=4096, spp=8192,
                    params={"max_depth": 32}, share=(0, 8)),
}


def config_scene(cfg):
    fn, args, kw = cfg["scene"]
    return getattr(scenes, fn)(*args, **kw)


def config_share_ids(cfg):
    """Pixel ids this GPU owns (None = the whole frame)."""
    if cfg["share"] is None:
        return None
    rank, world = cfg["share"]
    return scenes.tile_pixel_ids(cfg["width"], cfg["height"], rank, world)


def config_pixels(cfg):
    """The sparse subset the oracle renders: SUBSET pixels of the owned set, seeded choice, sorted."""
    own = config_share_ids(cfg)
    n = cfg["width"] * cfg["height"] if own is None else len(own)
    pick = np.sort(np.random.RandomState(20260203).choice(n, SUBSET, replace=False))
    return pick.astype(np.uint32) if own is None else np.ascontiguousarray(own[pick], np.uint32)


def config_params(cfg, spp, first_timestamp=0):
    p = abi.default_render_params(spp, first_timestamp)
    for k, v in cfg["params"].items():
        setattr(p, k, v)
    return p
