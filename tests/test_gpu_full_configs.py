"""The BASELINE.json configs at their STATED sizes on the GPU, through the C ABI.

Every frame is rendered at full resolution and the config's full sample count; a fixed sparse subset of its pixels
(256 per config) is compared bit for bit with what the CPU oracle computed for exactly those pixels and sample counts
(tests/golden/full_configs.npz, written by tests/golden/make_full_configs.py -- the oracle is not run here), the ray /
vertex counts of the subset must equal the oracle's, and size-independent properties are checked on the whole frame.

Tolerance (BASELINE.json north_star): per-pixel RMSE < 1e-3 at equal spp; asserted, and exact equality asserted on
top of it (same float32 operation order on both sides).
"""
import os
import zlib

import numpy as np
import pytest

from conftest import GOLDEN
from full_configs import CONFIGS, SPP_QUICK, config_params, config_pixels, config_scene, config_share_ids

pytestmark = pytest.mark.gpu

TOL_RMSE = 1e-3


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "full_configs.npz"))


def _rmse(a, b):
    return float(np.sqrt(((a[:, :3].astype(np.float64) - b[:, :3].astype(np.float64)) ** 2).mean()))


def _check_subset(img, where, expect, what):
    got = img[where]
    assert _rmse(got, expect) < TOL_RMSE, what
    bad = int((got != expect).any(1).sum())
    assert bad == 0, "%s: %d of %d oracle-checked pixels differ in some bit" % (what, bad, len(expect))


@pytest.mark.parametrize("name", list(CONFIGS))
def test_config_at_full_size(name, golden):
    import gpuspectral_amd as g

    cfg = CONFIGS[name]
    W, H, spp = cfg["width"], cfg["height"], cfg["spp"]
    sc = config_scene(cfg)
    own = config_share_ids(cfg)
    ids = config_pixels(cfg)
    assert np.array_equal(ids, golden[name + "/ids"]), "pixel subset changed: regenerate tests/golden/full_configs.npz"
    npix = W * H if own is None else len(own)
    where = ids.astype(np.int64) if own is None else np.searchsorted(own, ids)

    with g.Context(0) as ctx:
        ctx.upload_scene(sc)
        st0 = ctx.stats()
        assert st0["num_triangles"] == int(golden[name + "/triangles"][0])

        # ---- the frame at full resolution: first SPP_QUICK samples, twice (determinism), then on to the full count
        ctx.frame_begin(W, H, own)
        ctx.render(spp=SPP_QUICK, first_timestamp=0, params=config_params(cfg, SPP_QUICK))
        quick = ctx.download_compact()
        crc = zlib.crc32(quick.tobytes())
        _check_subset(quick, where, golden[name + "/quick"], "%s at %d spp" % (name, SPP_QUICK))
        ctx.frame_begin(W, H, own)
        ctx.reset_stats()
        ctx.render(spp=SPP_QUICK, first_timestamp=0, params=config_params(cfg, SPP_QUICK))
        again = ctx.download_compact()
        assert zlib.crc32(again.tobytes()) == crc, "two renders of the same frame differ"
        rest = spp - SPP_QUICK
        ctx.render(spp=rest, first_timestamp=SPP_QUICK, params=config_params(cfg, rest, SPP_QUICK))
        full = ctx.download_compact()
        st = ctx.stats()
        _check_subset(full, where, golden[name + "/full"], "%s at %d spp" % (name, spp))

        # ---- size-independent properties of the whole frame
        assert st["samples"] == npix * spp
        assert st["extension_rays"] >= st["samples"]                 # every sample shoots its primary ray
        assert st["shaded_vertices"] <= st["extension_rays"]         # one vertex per extension ray that hit
        assert st["shadow_rays"] <= st["shaded_vertices"]            # at most one NEE ray per vertex
        assert np.isfinite(full).all()
        assert (full[:, 3] == 1.0).all()                             # raygen.rgen:106 alpha
        assert (full[:, :3] >= 0.0).all()
        depth = cfg["params"].get("max_depth", 50)
        assert st["extension_rays"] <= st["samples"] * (depth + 2)   # raygen.rgen:73-75
        m = full[:, :3].mean()
        assert 1e-4 < m < 20.0, m                                     # lit, and below the firefly clamp (raygen.rgen:60)

        # ---- the subset alone: same pixels (any partition reproduces the frame) and the oracle's ray counts
        ctx.frame_begin(W, H, ids)
        ctx.reset_stats()
        ctx.render(spp=SPP_QUICK, first_timestamp=0, params=config_params(cfg, SPP_QUICK))
        sub = ctx.download_compact()
        s1 = ctx.stats()
        assert np.array_equal(sub, golden[name + "/quick"])
        got = [s1[k] for k in ("extension_rays", "shadow_rays", "shaded_vertices", "samples")]
        assert got == [int(v) for v in golden[name + "/counts_quick"]], (got, golden[name + "/counts_quick"])
        ctx.render(spp=rest, first_timestamp=SPP_QUICK, params=config_params(cfg, rest, SPP_QUICK))
        sub = ctx.download_compact()
        s2 = ctx.stats()
        assert np.array_equal(sub, golden[name + "/full"])
        got = [s2[k] for k in ("extension_rays", "shadow_rays", "shaded_vertices", "samples")]
        assert got == [int(v) for v in golden[name + "/counts_full"]], (got, golden[name + "/counts_full"])
        print("%s: %d tris %dx%d x %d spp on %d px: %.0f Mrays/s, %.0f Msamples/s" % (
            name, st["num_triangles"], W, H, spp, npix, st["traced_rays"] / st["render_seconds"] / 1e6,
            st["samples"] / st["render_seconds"] / 1e6))
