"""The product's per-path stage code (gpuspectral_amd/csrc/pt_*.h -- the functions the HIP
kernels call) compiled for the host by tests/emu, compared with the oracle.  Both sides are
float32 with the same operation order, so every comparison is bit-exact."""
import numpy as np
import pytest

from conftest import random_rays


def test_emu_render_cornell_bit_exact(emu, oracle_mod, cornell):
    img = emu.scene(cornell).render(64, 64, spp=4)
    ref, _ = oracle_mod.Oracle(cornell).render(64, 64, spp=4)
    assert np.array_equal(img, ref)


def test_emu_render_full_bsdf_set_bit_exact(emu, oracle_mod, materials_scene):
    img = emu.scene(materials_scene).render(48, 48, spp=6)
    ref, _ = oracle_mod.Oracle(materials_scene).render(48, 48, spp=6)
    assert np.array_equal(img, ref)
    assert np.isfinite(img).all()


def test_emu_nee_off_bit_exact(emu, oracle_mod, materials_scene, cornell):
    """RenderParams.nee = 0 (PathTracer.h:36-41; the `if (NEE)` / `!NEE ||` branches of rayhit.rchit:733,763-768): the
    product's shade_vertex against the oracle's closestHitShader, on the CPU.  No shadow ray is traced, every emitter met
    counts with full weight, and the light sample is still drawn -- so the paths are those of nee = 1 and only the
    estimator differs: same ray-segment count, no shadow rays, a different (noisier) image of the same scene."""
    from gpuspectral_amd import abi

    for sc, w, h, spp in ((materials_scene, 48, 48, 6), (cornell, 64, 64, 4)):
        p = abi.default_render_params()
        p.disable_nee = 1
        img = emu.scene(sc).render(w, h, spp=spp, params=p)
        ref, st = oracle_mod.Oracle(sc).render(w, h, spp=spp, params=p)
        assert np.array_equal(img, ref)
        on, st_on = oracle_mod.Oracle(sc).render(w, h, spp=spp)
        assert st["shadow_rays"] == 0 and st_on["shadow_rays"] > 0
        assert st["extension_rays"] == st_on["extension_rays"]  # the random streams coincide (rchit:720 is outside the branch)
        assert not np.array_equal(ref, on)
        # both estimators are of the same radiance: the frame means agree to Monte-Carlo accuracy
        assert abs(ref[:, :3].mean() - on[:, :3].mean()) < 0.35 * on[:, :3].mean()


@pytest.mark.parametrize("what", ["textures", "envmap", "both-srgb"])
def test_emu_dormant_features_bit_exact(emu, oracle_mod, what):
    """shade_vertex<true>, sample_texture, sample_envmap, det_atan2f of the product headers against the oracle's
    oracle_texture.h on the CPU (the GPU run of the same comparison is tests/test_gpu_textures.py)."""
    import textured

    sc = textured.decorate(textured.open_scene(10), seed=7, textures=what != "envmap", envmap=what != "textures",
                           decode=textured.srgb_table() if what == "both-srgb" else None)
    img = emu.scene(sc).render(64, 48, spp=3)
    ref, _ = oracle_mod.Oracle(sc).render(64, 48, spp=3)
    assert np.array_equal(img, ref)
    plain, _ = oracle_mod.Oracle(textured.open_scene(10)).render(64, 48, spp=3)
    assert not np.array_equal(ref, plain)


def test_emu_deep_paths_and_russian_roulette(emu, oracle_mod):
    """Dielectric-heavy scene: paths reach the RR regime (depth > 10) and the depth cap."""
    from gpuspectral_amd import abi, scenes

    sc = scenes.caustics(3000)
    p = abi.default_render_params()
    p.max_depth = 32
    img = emu.scene(sc).render(40, 40, spp=4, params=p)
    ref, st = oracle_mod.Oracle(sc).render(40, 40, spp=4, params=p)
    assert np.array_equal(img, ref)
    assert st["extension_rays"] / st["samples"] > 3


def test_emu_timestamps_and_subsets(emu, oracle_mod, cornell):
    e = emu.scene(cornell)
    a = e.render(32, 32, spp=2)
    a = e.render(32, 32, spp=3, first_timestamp=2, accum=a)
    ref, _ = oracle_mod.Oracle(cornell).render(32, 32, spp=5)
    assert np.array_equal(a, ref)
    ids = np.arange(3, 1024, 7, dtype=np.uint32)
    assert np.array_equal(e.render(32, 32, spp=5, pixel_ids=ids), ref[ids])


def test_emu_bsdf_tables_all_types(emu, oracle_mod, materials_scene):
    e = emu.scene(materials_scene)
    o = oracle_mod.Oracle(materials_scene)
    from gpuspectral_amd import abi

    rng = np.random.RandomState(5)
    n = 0
    for t, arr in enumerate(materials_scene.bsdfs):
        for i in range(len(arr)):
            h = abi.bsdf_handle(t, i)
            for _ in range(200):
                wo = rng.normal(size=3).astype(np.float32)
                wo /= np.linalg.norm(wo)
                seed = int(rng.randint(0, 2**31 - 1))
                a, b = e.bsdf_sample(h, wo, seed), o.bsdf_sample(h, wo, seed)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (t, i, wo, seed, a, b)
                wi = rng.normal(size=3).astype(np.float32)
                wi /= np.linalg.norm(wi)
                a, b = e.bsdf_eval(h, wo, wi), o.bsdf_eval(h, wo, wi)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (t, i, wo, wi, a, b)
                n += 1
    assert n >= 8 * 200
    # grazing / degenerate directions (NaN and inf must agree too)
    for wo in [(0, 0, 1), (1, 0, 0), (0, 1, 0), (0, 0, -1), (1e-20, 0, 1e-20)]:
        wo = np.array(wo, np.float32)
        for t, arr in enumerate(materials_scene.bsdfs):
            if len(arr):
                h = abi.bsdf_handle(t, 0)
                a, b = e.bsdf_sample(h, wo, 77), o.bsdf_sample(h, wo, 77)
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (t, wo)


def test_emu_light_sampling(emu, oracle_mod, materials_scene):
    e = emu.scene(materials_scene)
    o = oracle_mod.Oracle(materials_scene)
    rng = np.random.RandomState(9)
    for _ in range(500):
        pos = rng.uniform(-1, 1, 3).astype(np.float32) + np.array([0, 1, 0], np.float32)
        seed = int(rng.randint(0, 2**31 - 1))
        assert np.array_equal(e.sample_light(pos, seed).view(np.uint32), o.sample_light(pos, seed).view(np.uint32))


def test_emu_traversal_matches_oracle(emu, oracle_mod, materials_scene):
    """Closest hit is BVH-independent (min t, ties -> smaller triangle id): the product's
    traversal loop over a median-split tree equals the oracle over its SAH tree."""
    e = emu.scene(materials_scene)
    o = oracle_mod.Oracle(materials_scene)
    rays = random_rays(30000, 11)
    a, b = e.trace(rays), o.trace(rays)
    assert np.array_equal(a["prim"], b["prim"])
    hit = b["prim"] >= 0
    for k in ("t", "u", "v"):
        assert np.array_equal(a[k][hit], b[k][hit])
    rays[:, 3] = 0.01
    rays[:, 7] = np.random.RandomState(2).uniform(0.05, 3.0, len(rays))
    assert np.array_equal(e.trace(rays, True)["prim"], o.trace(rays, True)["prim"])


def test_emu_deterministic_math_and_inverse(emu, oracle_mod, cornell):
    x = np.random.RandomState(1).uniform(-90, 90, 50000).astype(np.float32)
    x = np.concatenate([x, np.array([0.0, -0.0, 1.0, np.inf, -np.inf, np.nan, 1e-40, 88.7, 88.8, -87.3, -87.4], np.float32)])
    s, c, lg, ex = (np.zeros_like(x) for _ in range(4))
    emu.L.emu_det_math(x.ctypes.data, x.size, s.ctypes.data, c.ctypes.data, lg.ctypes.data, ex.ctypes.data)
    rs, rc, rl, re = oracle_mod.det_math(x)
    for a, b in ((s, rs), (c, rc), (lg, rl), (ex, re)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    for inst in cornell.instances:
        m = np.ascontiguousarray(inst["transform"])
        out = np.zeros(16, np.float32)
        emu.L.emu_transform_inv_t(m.ctypes.data, out.ctypes.data)
        assert np.array_equal(out, oracle_mod.transform_inv_t(m))
    assert emu.L.emu_seed(128, 5, 9, 3) == oracle_mod.pcg_hash(oracle_mod.tea(128 * 9 + 5, 3))


def test_leaf_step_triangle_test_equals_the_oracle_statement(emu):
    """k_trace's leaf step runs the watertight test as straight-line code with the ray's axis permutation applied through two
    lane masks and without make_shear's kx / ky exchange (pt_trace.h intersect_tri_rot).  Same verdict and the same t, u, v bit
    for bit as the oracle's statement (intersect_tri) -- for rays of every dominant axis and sign, rays through shared edges and
    vertices (the double-precision fallback), axis-parallel rays, degenerate and far-away triangles."""
    rng = np.random.RandomState(77)
    n = 400000
    tris = rng.uniform(-2, 2, (n, 9)).astype(np.float32)
    tris[: n // 8] *= np.float32(1e-3)  # tiny
    tris[n // 8: n // 4] += np.float32(1e4)  # far from the origin
    o = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    k = rng.randint(0, 4, n)
    bary = rng.dirichlet((1, 1, 1), n).astype(np.float32)
    bary[k == 1] = np.eye(3, dtype=np.float32)[rng.randint(0, 3, (k == 1).sum())]  # through a vertex
    e = k == 2
    bary[e, 2] = 0
    bary[e, 0] = rng.uniform(0, 1, e.sum()).astype(np.float32)
    bary[e, 1] = np.float32(1) - bary[e, 0]  # through an edge
    tgt = (tris.reshape(n, 3, 3) * bary[:, :, None]).sum(1)
    tgt[k == 3] = rng.uniform(-3, 3, ((k == 3).sum(), 3)).astype(np.float32)  # anywhere: mostly misses
    d = tgt - o
    ax = rng.randint(0, 12, n)
    for a in range(3):  # axis-parallel rays (two zero components), both signs
        m = ax == a
        z = np.zeros((m.sum(), 3), np.float32)
        z[:, a] = np.where(rng.rand(m.sum()) < 0.5, -1, 1)
        d[m] = z
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-30).astype(np.float32)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3], rays[:, 3:6] = o, d
    rays[:, 6] = np.where(rng.rand(n) < 0.5, 0.0, 0.01)
    rays[:, 7] = np.where(rng.rand(n) < 0.5, 1e10, rng.uniform(0.1, 8, n))
    tris[-1000:, 6:9] = tris[-1000:, 0:3]  # zero-area triangles
    a, b = np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)
    rays, tris = np.ascontiguousarray(rays), np.ascontiguousarray(tris)
    emu.L.emu_tri_tests(rays.ctypes.data, tris.ctypes.data, n, a.ctypes.data, b.ctypes.data)
    assert a[:, 0].sum() > n // 5  # (the case set does hit)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_node_encoder_forms_agree(emu):
    """pt_trace.h states the 4-wide node encoder twice: encode_node_w4_ref (loops over the children, arrays indexed by loop
    variables -- the definition) and encode_node_w4 (what the device build and refit run: child positions in registers, the stable
    insertion sort of the per-octant keys as six chained compare-exchanges, the order code in closed form, power-of-two divisions
    as multiplications).  Same record, bit for bit: every child count, boxes of every size from 1e-30 to 1e30, children that
    coincide (ties in every octant), flat and point-like nodes, huge offsets from the origin."""
    rng = np.random.RandomState(20250)
    n = 400_000
    centre = rng.uniform(-1, 1, (n, 1, 3)) * 10.0 ** rng.uniform(-3, 6, (n, 1, 1))
    size = 10.0 ** rng.uniform(-30, 30, (n, 1, 1)) * (rng.rand(n, 1, 1) < 0.1) + 10.0 ** rng.uniform(-4, 3, (n, 1, 1)) * 1.0
    size = np.minimum(size, 1e30)
    lo = centre + rng.uniform(-1, 1, (n, 4, 3)) * size
    ext = rng.uniform(0, 1, (n, 4, 3)) * size * (rng.rand(n, 4, 3) > 0.1)  # flat children on some axes
    hi = lo + ext
    boxes = np.concatenate([lo, hi], axis=2).astype(np.float32)
    dup = rng.rand(n) < 0.2  # ties: children 1.. repeat child 0 (or each other)
    boxes[dup, 1] = boxes[dup, 0]
    dup2 = rng.rand(n) < 0.1
    boxes[dup2, 3] = boxes[dup2, 2]
    snap = rng.rand(n) < 0.1  # centres on a coarse grid: equal keys in some octants only
    boxes[snap] = np.round(boxes[snap] * 4) / 4
    boxes[snap, :, 3:] = np.maximum(boxes[snap, :, 3:], boxes[snap, :, :3])
    cnt = rng.randint(0, 5, n)
    ni = (rng.rand(n) * (cnt + 1)).astype(np.int32)
    counts = np.stack([ni, cnt - ni], axis=1).astype(np.int32)
    boxes, counts = np.ascontiguousarray(boxes), np.ascontiguousarray(counts)
    a, b = np.zeros((n, 16), np.uint32), np.zeros((n, 16), np.uint32)
    emu.L.emu_encode_nodes(boxes.ctypes.data, counts.ctypes.data, n, a.ctypes.data, b.ctypes.data)
    assert len(np.unique(a[:, 12])) > 1000  # (many different order words)
    bad = np.nonzero((a != b).any(1))[0]
    assert len(bad) == 0, (len(bad), counts[bad[:5]], boxes[bad[:2]], a[bad[:2]], b[bad[:2]])
