"""Pins of the CPU oracle: the RNG known answers and worked numbers SURVEY.md derived
from the reference sources, the frozen golden fixtures, and the one image fixture the
reference tree holds (Tungsten ground truth of the classic Cornell box)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN


def test_rng_known_answers(oracle_mod):
    """values computed by hand from pt_common.glsl:86-120."""
    o = oracle_mod
    kat = [
        ((0, 0), 0x5DF5F2BF, 0x66F8BB0C, (0xE4737E43, 0x8A69CF8E, 0xAB953492)),
        ((1, 0), 0xC09848F2, 0x3A910142, (0x223461C8, 0x0739263E, 0x4F8DD381)),
        ((0, 1), 0x5D8714D7, 0xC9A3BA20, (0xB168C2CE, 0xF2630107, 0xAFAC66A3)),
    ]
    for (a, b), t, h, outs in kat:
        assert o.tea(a, b) == t
        assert o.pcg_hash(t) == h
        got, _ = o.rand_pcg(h, 3)
        assert tuple(int(x) for x in got) == outs


def test_rand_uniform_is_inclusive_of_one(oracle_mod):
    """float(0xffffffffu) rounds to 2^32: randUniform() can return exactly 1.0 (pt_common.glsl:102-104)."""
    # find a state whose output word is >= 0xffffff80 by brute force over a short orbit is impractical;
    # check the scale instead: uniform == float(word) * 2^-32 for the first words of a stream
    words, _ = oracle_mod.rand_pcg(12345, 64)
    state = 12345
    for w in words[:8]:
        u = oracle_mod.rand_uniform(state)
        assert u == np.float32(np.float32(w) * np.float32(2.0 ** -32))
        state = (state * 747796405 + 2891336453) & 0xFFFFFFFF
    assert np.float32(np.float32(0xFFFFFFFF) * np.float32(2.0 ** -32)) == np.float32(1.0)


def test_cornell_camera_rays(oracle_mod, cornell):
    """primary directions of the Cornell camera at 128x128."""
    o = oracle_mod.Oracle(cornell)
    exp = {
        (64, 64): (0.0, 0.0, -1.0),
        (0, 0): (-0.166972, 0.166972, -0.971720),
        (127, 127): (0.164505, -0.164505, -0.972562),
        (64, 0): (0.0, 0.169350, -0.985556),
    }
    for (x, y), d in exp.items():
        r = o.primary_ray(128, 128, x, y)
        assert np.allclose(r[:3], (0.0, 1.0, 6.8), atol=1e-6)
        assert np.allclose(r[3:], d, atol=2e-6), (x, y, r)
    hit = o.trace(np.array([[0, 1.5, 6.8, 0, 0, 0, -1, 1e10]], np.float32))[0]
    assert abs(hit["t"] - 7.8) < 1e-5  # above the tall box: the back wall z = -1 (instance 2)
    assert hit["prim"] in (4, 5)


def test_cornell_scene_structure(cornell):
    """8 instances, 36 triangles, 2 lights of area 0.0893."""
    assert len(cornell.instances) == 8 and cornell.num_triangles == 36 and len(cornell.lights) == 2
    assert (cornell.instances["twofaced"] == 1).all()
    assert (cornell.instances["bsdf"] >> 16 == 0).all()  # diffuse only
    assert np.allclose(cornell.instances["emission"][7], (17, 12, 4))
    p = cornell.lights["positions"][0][:, :3].astype(np.float64)
    assert np.allclose(p, [(-0.24, 1.98, 0.16), (-0.24, 1.98, -0.22), (0.23, 1.98, 0.16)], atol=1e-6)
    area = 0.5 * np.linalg.norm(np.cross(p[1] - p[0], p[2] - p[0]))
    assert abs(area - 0.0893) < 1e-6
    assert abs(float(cornell.fov) - 0.3403392) < 1e-7


def test_deterministic_transcendentals_accuracy(oracle_mod):
    """The oracle's sin/cos/log/exp stay within a few ulp of libm on the ranges the shaders use."""
    rng = np.random.RandomState(0)
    x = np.concatenate([rng.uniform(-np.pi / 4, 2 * np.pi + 0.1, 20000), np.linspace(-0.8, 6.4, 5000)]).astype(np.float32)
    s, c, _, _ = oracle_mod.det_math(x)
    ulp = lambda a, ref: np.abs(a.astype(np.float64) - ref) / np.maximum(np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64), 1e-45)
    # near the zeros of sin/cos the error is absolute (GLSL's own bound is 2^-11 absolute)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 3e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 3e-7
    xl = rng.uniform(2.0 ** -24, 1.0, 20000).astype(np.float32)
    _, _, lg, _ = oracle_mod.det_math(xl)
    assert ulp(lg, np.log(xl.astype(np.float64))).max() < 4
    xe = rng.uniform(-80.0, 0.0, 20000).astype(np.float32)
    _, _, _, ex = oracle_mod.det_math(xe)
    assert ulp(ex, np.exp(xe.astype(np.float64))).max() < 4
    _, _, lg0, ex0 = oracle_mod.det_math(np.array([0.0, 1.0, -np.inf, -200.0], np.float32))
    assert lg0[0] == -np.inf and lg0[1] == 0.0 and ex0[2] == 0.0 and ex0[3] == 0.0 and ex0[0] == 1.0


def test_golden_cornell_image(oracle_mod, cornell):
    """BASELINE config 1 (Cornell 128x128, 1 spp, timestamp 0): frozen oracle output."""
    img, st = oracle_mod.Oracle(cornell).render(128, 128, spp=1)
    gold = np.load(os.path.join(GOLDEN, "cornell_128_1spp.npy")).reshape(-1, 3)
    assert np.array_equal(img[:, :3], gold)
    assert (img[:, 3] == 1.0).all()
    assert st["samples"] == 128 * 128 and st["extension_rays"] > st["samples"]


def test_golden_first_hit_map(oracle_mod, cornell):
    o = oracle_mod.Oracle(cornell)
    g = np.load(os.path.join(GOLDEN, "cornell_first_hit_128.npz"))
    rays = np.zeros((128 * 128, 8), np.float32)
    for y in range(0, 128):
        for x in range(128):
            r = o.primary_ray(128, 128, x, y)
            rays[y * 128 + x, 0:3] = r[:3]
            rays[y * 128 + x, 4:7] = r[3:]
    rays[:, 7] = 1e10
    hits = o.trace(rays)
    assert np.array_equal(hits["prim"], g["prim"]) and np.array_equal(hits["t"], g["t"])
    # closed box: primary rays hit, except the exact-diagonal pixels that aim at the hairline seam
    # between wall rectangles (their matrices carry 1e-8 noise, scene.xml:68-94)
    assert (hits["prim"] >= 0).mean() > 0.998


def test_golden_bsdf_vectors(oracle_mod):
    """Per-BSDF sample/eval and light-sampling tables for fixed (wo, seed) inputs, all 8 types."""
    from gpuspectral_amd import scenes

    o = oracle_mod.Oracle(scenes.cornell_materials(8))
    g = np.load(os.path.join(GOLDEN, "bsdf_vectors.npz"))
    types = set()
    for i, (h, wo, s) in enumerate(zip(g["handles"], g["wo"], g["seeds"])):
        smp = o.bsdf_sample(int(h), wo, int(s))
        assert np.array_equal(smp.view(np.uint32), g["samples"][i]), (i, hex(int(h)))
        ev = o.bsdf_eval(int(h), wo, smp[:3])
        assert np.array_equal(ev.view(np.uint32), g["evals"][i]), (i, hex(int(h)))
        types.add(int(h) >> 16)
    assert types == set(range(8))
    for i in range(len(g["lights"])):
        ls = o.sample_light(g["wo"][i] * 0.5 + np.array([0, 1, 0], np.float32), int(g["seeds"][i]))
        assert np.array_equal(ls.view(np.uint32), g["lights"][i])


def test_golden_materials_image(oracle_mod, materials_scene):
    from gpuspectral_amd import scenes

    img, _ = oracle_mod.Oracle(scenes.cornell_materials(8)).render(64, 64, spp=4)
    gold = np.load(os.path.join(GOLDEN, "materials_64_4spp.npy")).reshape(-1, 3)
    assert np.array_equal(img[:, :3], gold)


def test_tungsten_ground_truth(oracle_mod, cornell):
    """The reference-held image fixture for the Cornell box (cornell-box/TungstenRender.png, a third-party ground truth
    linearised by tests/golden/make_tungsten.py) against the CPU oracle, cell by cell like the GPU test
    (tests/test_gpu_reference_images.py, which renders 1024 x 1024 x 1024 spp).  Here 128 x 128 x 64 spp (one ray
    direction per pixel, no filter: the cells carry ~3 % of noise and aliasing).  Measured: 146 cell-channels, ratio
    0.947 .. 1.323, median 1.029; lit walls 0.964 .. 1.033.  The reference integrator is biased (MIS quirks, firefly
    clamp: SURVEY 8a13), so this pins the oracle to ~5 %, not to the bit."""
    from test_gpu_reference_images import REF, _cells

    fix = np.load(os.path.join(REF, "tungsten_cornell-box.npz"))
    img, _ = oracle_mod.Oracle(cornell).render(128, 128, spp=64)
    ours = img[:, :3].reshape(128, 128, 3).astype(np.float64)
    r, _ = _cells(ours, fix)
    assert np.isfinite(r).sum() >= 140
    assert 0.90 < np.nanmin(r) and np.nanmax(r) < 1.40, (np.nanmin(r), np.nanmax(r))
    assert 0.99 < np.nanmedian(r) < 1.07
    walls = r[1:3, 1:7]
    assert 0.93 < np.nanmin(walls) and np.nanmax(walls) < 1.07
    # red wall left, green wall right
    assert ours[64, 4, 0] > 3 * ours[64, 4, 1] and ours[64, 123, 1] > 2 * ours[64, 123, 0]


def test_bias_attribution_against_tungsten(oracle_mod, cornell):
    """The excess of the reference's estimator over the unbiased third-party render is MEASURED, not asserted.
    oracle_set_quirks_off (oracle/oracle_bsdf.h, kQuirk*) replaces the reference's four documented departures from a
    textbook MIS path tracer, one bit each, by their textbook forms.  Cornell box, 384 x 384 x 64 spp, strict cell mask
    (tests/test_gpu_reference_images.py::_cells: no shoulder / toe / silhouette blocks), 123 cell-channels.  Measured
    (profiles/r06_quirk_attribution.txt at 512 x 512 x 128 spp; here within noise of it):
        as shipped (mask 0)        1.003 .. 1.289, median 1.026   -- tall box and its shadow 1.06 .. 1.29
        all four off (mask 15)     0.998 .. 1.028, median 1.010   -- EVERY cell, tall box and shadow included
        only bit 2 off             0.999 .. 1.029                 -- i.e. the whole excess is the emitter-hit weight
                                                                     (rayhit.rchit:763-765,785-790: the pdf of the previous
                                                                     vertex's light SAMPLE, 1 when that sample was shadowed)
        only bit 1 / only bit 4    change < 0.1 % / nothing on this scene (a small light: the NEE weight is ~1 either way;
                                                                     no contribution reaches the cutoff of 20)
    So a misreading of a pdf shared by oracle and kernels would have to hide inside +-3 % of an unbiased renderer on every
    cell, not inside the 0.98 .. 1.35 band the as-shipped comparison allows.  The remaining +1 % is common to all cells
    (8-bit tone-mapped fixture, Tungsten's tent filter)."""
    from test_gpu_reference_images import REF, _cells

    fix = np.load(os.path.join(REF, "tungsten_cornell-box.npz"))
    L = oracle_mod.lib()

    def cells(mask):
        L.oracle_set_quirks_off(mask)
        try:
            img, _ = oracle_mod.Oracle(cornell).render(384, 384, spp=64)
        finally:
            L.oracle_set_quirks_off(0)
        ours = img[:, :3].astype(np.float64).reshape(128, 3, 128, 3, 3).mean(axis=(1, 3))
        return _cells(ours, fix, strict=True)[0]

    shipped, textbook, only2 = cells(0), cells(15), cells(2)
    ok = np.isfinite(textbook)
    assert ok.sum() >= 115 and np.array_equal(ok, np.isfinite(shipped))
    # all four departures off: an unbiased estimator, and it agrees with the unbiased renderer on every cell
    assert 0.975 < textbook[ok].min() and textbook[ok].max() < 1.035, (textbook[ok].min(), textbook[ok].max())
    med = np.median(textbook[ok])
    assert 1.0 < med < 1.02 and np.abs(textbook[ok] / med - 1.0).max() < 0.025, (med, np.abs(textbook[ok] / med - 1.0).max())
    # as shipped: today's band reappears, largest on the tall box and in its shadow (rows 3-7, columns 0-3)
    assert shipped[ok].max() > 1.18 and shipped[ok].min() > 0.975 and 1.015 < np.median(shipped[ok]) < 1.04
    box = np.zeros_like(ok)
    box[3:8, 0:4] = True
    assert np.nanmax(shipped[ok & box]) == shipped[ok].max() and np.nanmax(shipped[ok & ~box]) < 1.13
    # ... and it is the emitter-hit weight alone
    assert np.abs(only2[ok] / textbook[ok] - 1.0).max() < 0.01


def test_quirk_mask_zero_is_the_default_and_changes_nothing(oracle_mod, cornell):
    L = oracle_mod.lib()
    L.oracle_get_quirks_off.restype = C.c_uint32
    assert L.oracle_get_quirks_off() == 0
    a, _ = oracle_mod.Oracle(cornell).render(32, 32, spp=3)
    L.oracle_set_quirks_off(15)
    b, _ = oracle_mod.Oracle(cornell).render(32, 32, spp=3)
    L.oracle_set_quirks_off(0)
    c, _ = oracle_mod.Oracle(cornell).render(32, 32, spp=3)
    assert np.array_equal(a, c) and not np.array_equal(a, b)


def test_running_mean_and_subsets(oracle_mod, cornell):
    """mix() recurrence (raygen.rgen:84-108): split renders equal one render; pixel subsets equal the frame."""
    o = oracle_mod.Oracle(cornell)
    a, _ = o.render(32, 32, spp=5)
    b, _ = o.render(32, 32, spp=2)
    b, _ = o.render(32, 32, spp=3, first_timestamp=2, accum=b)
    assert np.array_equal(a, b)
    ids = np.arange(7, 32 * 32, 5, dtype=np.uint32)
    c, _ = o.render(32, 32, spp=5, pixel_ids=ids)
    assert np.array_equal(c, a[ids])
    one, _ = o.render(32, 32, spp=5, threads=1)
    assert np.array_equal(one, a)  # thread count does not change results


def test_transform_inv_t_is_inverse_transpose(oracle_mod, cornell):
    for inst in cornell.instances:
        m = inst["transform"]
        it = oracle_mod.transform_inv_t(m).reshape(4, 4).T.astype(np.float64)  # math layout
        M = m.reshape(4, 4).T.astype(np.float64)
        assert np.allclose(it, np.linalg.inv(M.T), rtol=1e-5, atol=1e-6)
