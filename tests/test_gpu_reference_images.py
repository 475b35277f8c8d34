"""The HIP path against the only expected OUTPUTS the reference tree holds: the third-party Tungsten ground-truth
renders shipped beside its scenes (tests/golden/ref_scenes/tungsten_*.npz, linearised by tests/golden/make_tungsten.py).

This is independent of the CPU oracle: nothing under oracle/ is used here.  It cannot be a bit-parity test -- Tungsten
is an unbiased path tracer with a tent pixel filter, the reference integrator has its documented MIS quirks and a
firefly clamp, and the PNGs are 8-bit tone-mapped -- so every check states the band it allows and where
the numbers it was chosen from are (profiles/r02_tungsten_compare.txt, measured at 1024 spp).

Frames are rendered at the scene's own film size, then box-filtered 8x8 like the fixtures and compared on cells of
16x16 such blocks (128x128 pixels); a cell takes part per channel when at least a quarter of its blocks are
recoverable from the PNG (not clipped, not in the tone map's toe) and brighter than 0.02.
"""
import os

import numpy as np
import pytest

from conftest import CORNELL_XML, GOLDEN

pytestmark = pytest.mark.gpu

REF = os.path.join(GOLDEN, "ref_scenes")
SPP = 1024
CELL = 16


def _render(sc, W, H, spp=SPP):
    import gpuspectral_amd as g

    with g.Context(0) as ctx:
        ctx.upload_scene(sc)
        ctx.frame_begin(W, H)
        ctx.render(spp=spp)
        img = ctx.download()
    assert np.isfinite(img).all()
    return img[..., :3].astype(np.float64).reshape(H // 8, 8, W // 8, 8, 3).mean(axis=(1, 3))


def _cells(ours, fix, strict=False):
    """Per-cell, per-channel ratio ours / Tungsten over the recoverable blocks (NaN where too few).
    strict: also leave out the blocks on the tone curve's shoulder (fixture key `shoulder`, PNG level >= 232: one
    8-bit level there is 5-50 % of radiance -- the light source itself, which `sat` lets through at level 252)."""
    t = fix["lin"].astype(np.float64)
    ok = ~fix["sat"] & (t > 0.02)
    if strict:
        # what a comparison to a few per cent may use: no block next to the shoulder (the light and its antialiased rim),
        # nothing below PNG level ~40 (radiance 0.03: one level is > 3 % there), and no block on a silhouette -- the two
        # renderers' pixel grids are a pixel apart (raygen.rgen:21 has no half-pixel offset, Tungsten filters with a
        # tent), which moves a block mean by the edge contrast / 8
        def grow(m):
            p = np.pad(m, ((1, 1), (1, 1), (0, 0)), mode="edge")
            return np.stack([p[i:i + m.shape[0], j:j + m.shape[1]] for i in range(3) for j in range(3)]).any(axis=0)

        p = np.pad(t, ((1, 1), (1, 1), (0, 0)), mode="edge")
        nb = np.stack([p[i:i + t.shape[0], j:j + t.shape[1]] for i in range(3) for j in range(3)])
        edge = nb.max(axis=0) > 1.25 * nb.min(axis=0)
        ok &= ~grow(fix["shoulder"].any(axis=2, keepdims=True) | np.zeros_like(fix["sat"])) & (t > 0.03) & ~edge
    h, w = t.shape[:2]
    hc, wc = h // CELL, w // CELL
    cut = (slice(0, hc * CELL), slice(0, wc * CELL))
    shp = (hc, CELL, wc, CELL, 3)
    ts = np.where(ok, t, 0.0)[cut].reshape(shp).sum(axis=(1, 3))
    os_ = np.where(ok, ours, 0.0)[cut].reshape(shp).sum(axis=(1, 3))
    n = ok[cut].reshape(shp).sum(axis=(1, 3))
    return np.where(n >= CELL * CELL // 4, os_ / np.maximum(ts, 1e-12), np.nan), ok


def test_cornell_box_against_tungsten():
    """Diffuse-only scene, square film, level camera: every recoverable cell of the image takes part.
    Measured (1024 spp): 146 cell-channels, ratio 1.011 .. 1.299, median 1.031; the lit walls (rows 1-2) 1.011 .. 1.028.
    The excess over 1 is the reference integrator's own bias and is reproduced, not corrected.  Which bias, measured on the
    CPU oracle by switching the reference's four departures from a textbook MIS estimator off one at a time
    (tests/test_oracle_pins.py::test_bias_attribution_against_tungsten, profiles/r06_quirk_attribution.txt): all of it is the
    emitter-hit weight of rayhit.rchit:763-765,785-790 (the pdf of the previous vertex's light SAMPLE instead of the pdf of
    reaching the hit point, and 1 when that sample was shadowed -- largest on the tall box and in its shadow, 1.06 .. 1.29);
    the NEE weight of :751 changes < 0.1 % on this scene and the cutoff nothing.  With the four off the oracle is within
    0.998 .. 1.028 of Tungsten on every cell.
    Excluded by the fixture's mask: the light source (clipped in the PNG) and the near-black short box front."""
    from gpuspectral_amd import host

    sc = host.Scene(CORNELL_XML, os.path.dirname(os.path.dirname(CORNELL_XML))).arrays()
    ours = _render(sc, 1024, 1024)
    fix = np.load(os.path.join(REF, "tungsten_cornell-box.npz"))
    r, _ = _cells(ours, fix)
    valid = np.isfinite(r)
    assert valid.sum() >= 140
    assert 0.98 < np.nanmin(r) and np.nanmax(r) < 1.35, (np.nanmin(r), np.nanmax(r))
    assert 1.00 < np.nanmedian(r) < 1.06, np.nanmedian(r)
    walls = r[1:3, 1:7]  # back wall + side walls at mid height: direct light dominates
    assert np.isfinite(walls).sum() >= 30
    assert 0.995 < np.nanmin(walls) and np.nanmax(walls) < 1.04, (np.nanmin(walls), np.nanmax(walls))
    # hue of the walls: red left, green right (absolute radiance, both images)
    t = fix["lin"]
    for img in (ours, t):
        assert img[64, 4, 0] > 5 * img[64, 4, 1] and img[64, 123, 1] > 2 * img[64, 123, 0]


def test_staircase2_against_tungsten(staircase2_xml):
    """'Modern Hall' (30 927 triangles, 16 emitters, <ref>/twosided/roughplastic materials).  Its floor and stair treads
    are *textured* in Tungsten; the reference never sets hasTexture (rayhit.rchit:716,729; Loader.cpp:122-143 is dead
    code), so they render with the default colour and everything lit through them (the left corridor, the lower half
    of the frame) is expected to differ and is excluded.  Compared: the plain right-hand wall, rows 0-3 x columns 4-7
    of the cell grid.  Measured (1024 spp): ratio 0.88 .. 1.40, median 1.13 (the missing floor bounce and the clamp
    move light around), wall chromaticity (0.45, 0.36, 0.19) vs Tungsten's (0.46, 0.35, 0.18)."""
    from gpuspectral_amd import abi, host

    # SURVEY 8(f).1 on the GPU box: the reference's own scene.xml + OBJ files through the product's C++ loader
    sc = host.Scene(staircase2_xml).arrays()
    cached = abi.SceneArrays.load(os.path.join(REF, "staircase2.npz"))
    assert sc.num_triangles == 30927 and np.array_equal(sc.positions, cached.positions) and np.array_equal(sc.instances, cached.instances)
    ours = _render(sc, 1024, 1024)
    fix = np.load(os.path.join(REF, "tungsten_staircase2.npz"))
    r, ok = _cells(ours, fix)
    wall = r[0:4, 4:8]
    assert np.isfinite(wall).all()
    assert 0.75 < wall.min() and wall.max() < 1.55, (wall.min(), wall.max())
    assert 1.0 < np.median(wall) < 1.25, np.median(wall)
    t = fix["lin"].astype(np.float64)
    region = (slice(0, 64), slice(64, 128))
    ct = np.where(ok, t, 0.0)[region].sum(axis=(0, 1))
    co = np.where(ok, ours, 0.0)[region].sum(axis=(0, 1))
    assert np.abs(ct / ct.sum() - co / co.sum()).max() < 0.03


def test_staircase2_with_dormant_features_against_tungsten(staircase2_xml):
    """SURVEY 8(f).3 checked against a fixture the reference holds: the same scene loaded with
    LoadOptions::dormantFeatures (wood5.jpg / Tiles.jpg / Wallpaper.jpg on the treads, the floor and the back wall, read by
    the product's own JPEG decoder, sRGB-decoded) must agree with Tungsten's render where the reference's way of loading
    it cannot: the textured floor and everything lit through it.  Measured at 1024 spp (profiles/r02_texture_probe.txt),
    ratio ours / Tungsten per cell-channel, reference behaviour -> dormant features:
      whole frame   median 0.64 -> 1.06, mean |log2 ratio| 1.16 -> 0.46
      lower half    median 0.37 -> 1.01, mean |log2 ratio| 1.72 -> 0.39     (rows 4-7: floor + stairs)
      bottom rows   0.22 .. 0.60 -> 0.92 .. 1.24                            (rows 6-7 without column 1: the tiled floor itself)
    and with the bytes taken as linear values (srgb_textures = False) the floor comes out 1.4-2.2 x too bright: the
    sRGB table is what the files mean.  What stays outside: one column of cells (1, rows 2-6) that looks at geometry
    the loader skips (sphere shapes, S/engine/Loader.cpp:276-283), and the integrator's own excess on directly lit
    walls (1.1-1.4, as in the untextured test above)."""
    from gpuspectral_amd import host

    fix = np.load(os.path.join(REF, "tungsten_staircase2.npz"))

    def cells(**kw):
        scene = host.Scene(staircase2_xml, **kw)
        r, _ = _cells(_render(scene.arrays(), 1024, 1024), fix)
        return r, scene

    plain, _ = cells()
    tex, scene = cells(dormant_features=True)
    assert len(scene.arrays().textures) == 3 and not [w for w in scene.warnings if "textured" in w]

    def spread(r):
        return float(np.nanmean(np.abs(np.log2(r[np.isfinite(r)]))))

    assert np.isfinite(tex).sum() >= 180
    assert 0.95 < np.nanmedian(tex) < 1.15 and np.nanmedian(plain) < 0.75, (np.nanmedian(tex), np.nanmedian(plain))
    assert spread(tex) < 0.6 and spread(plain) > 1.0 and spread(tex[4:8]) < 0.5 and spread(plain[4:8]) > 1.4
    floor = tex[6:8][:, [0, 2, 3, 4, 5, 6, 7]]  # (column 1: see above)
    assert np.isfinite(floor).all() and 0.85 < floor.min() and floor.max() < 1.3, (floor.min(), floor.max())
    assert np.nanmax(plain[6:8]) < 0.7
    # hue of the floor tiles: the textured render's chromaticity is Tungsten's, cell by cell
    chroma = floor / floor.mean(axis=2, keepdims=True)
    assert np.abs(chroma - 1.0).max() < 0.12, np.abs(chroma - 1.0).max()


def test_coffee_against_tungsten(coffee_xml):
    """'Coffee Maker' (168 199 triangles; smooth/rough plastic, dielectric glass, rough conductor).  The camera is pitched
    and the reference flips the ray's *world-space* y (raygen.rgen:25: d = toWorld * d; d.y *= -1), which displaces
    the image vertically against Tungsten's (measured: ~14 of 125 block rows), and the film is portrait while the
    reference scales by max(W, H) (raygen.rgen:22).  So positions do not line up and the comparison is by material,
    not by place: mean radiance of the orange plastic body (mask: r > 2 b + 0.05, 0.15 < r < 0.95) and of the grey
    backdrop beside it.  Measured (1024 spp): body (0.640, 0.137, 0.031) vs Tungsten (0.620, 0.150, 0.039); backdrop
    0.045 / 0.034 vs 0.056 / 0.041 (left / right edge, mid height)."""
    from gpuspectral_amd import abi, host

    # SURVEY 8(f).1 on the GPU box: the reference's own scene.xml + OBJ files through the product's C++ loader (r03; the
    # flattened copy committed in r01 must be what it produces)
    sc = host.Scene(coffee_xml).arrays()
    cached = abi.SceneArrays.load(os.path.join(REF, "coffee.npz"))
    assert sc.num_triangles == 168199 and np.array_equal(sc.positions, cached.positions) and np.array_equal(sc.instances, cached.instances)
    assert all(np.array_equal(a, b) for a, b in zip(sc.bsdfs, cached.bsdfs)) and np.array_equal(sc.lights, cached.lights)
    ours = _render(sc, 800, 1000)
    t = np.load(os.path.join(REF, "tungsten_coffee.npz"))["lin"].astype(np.float64)

    def body(a):
        m = (a[..., 0] > 2.0 * a[..., 2] + 0.05) & (a[..., 0] > 0.15) & (a[..., 0] < 0.95)
        assert m.sum() > 800
        return a[m].mean(axis=0)

    bo, bt = body(ours), body(t)
    assert 0.9 < bo[0] / bt[0] < 1.15 and 0.75 < bo[1] / bt[1] < 1.15 and 0.6 < bo[2] / bt[2] < 1.2, (bo, bt)
    for cols in (slice(2, 8), slice(92, 98)):
        ro = ours[60:70, cols].mean() / t[60:70, cols].mean()
        assert 0.65 < ro < 1.1, ro
