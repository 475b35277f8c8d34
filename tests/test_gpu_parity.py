"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Arithmetic is float32 end to end; both sides use the same operation order and
the same deterministic transcendentals, so images are expected to agree bit for
bit.  The contract (BASELINE.json north_star) is per-pixel RMSE < 1e-3 at equal
spp; the tests assert that bound and report exactness separately.
"""
import os

import numpy as np
import pytest

from conftest import random_rays, rmse

pytestmark = pytest.mark.gpu

TOL_RMSE = 1e-3  # BASELINE.json: "per-pixel RMSE vs reference < 1e-3"


@pytest.fixture(scope="module", params=["default", "wavefront-only"])
def ctx(request):
    """Every test that takes `ctx` runs twice: with the defaults (frames this small finish in k_finish, one path per
    lane) and with GSP_FINISH_PATHS=0, which keeps every bounce in the wavefront kernels the bench measures."""
    import os

    import gpuspectral_amd as g

    old = os.environ.get("GSP_FINISH_PATHS")
    if request.param == "wavefront-only":
        os.environ["GSP_FINISH_PATHS"] = "0"
    try:
        c = g.Context(0)
    finally:
        if old is None:
            os.environ.pop("GSP_FINISH_PATHS", None)
        else:
            os.environ["GSP_FINISH_PATHS"] = old
    yield c
    c.close()


def primary_rays(orc_scene, W, H):
    rays = np.zeros((W * H, 8), np.float32)
    for y in range(H):
        for x in range(W):
            r = orc_scene.primary_ray(W, H, x, y)
            rays[y * W + x, 0:3] = r[0:3]
            rays[y * W + x, 4:7] = r[3:6]
    rays[:, 3] = 0.0
    rays[:, 7] = 1e10
    return rays


@pytest.mark.parametrize("which", ["cornell", "materials"])
def test_trace_closest_hit_matches_oracle(ctx, oracle_mod, cornell, materials_scene, which):
    sc = cornell if which == "cornell" else materials_scene
    o = oracle_mod.Oracle(sc)
    ctx.upload_scene(sc)
    rays = np.concatenate([primary_rays(o, 64, 64), random_rays(20000, 3)])
    got = ctx.trace(rays)
    ref = o.trace(rays)
    assert (got["prim"] == ref["prim"]).all(), "hit triangle differs on %d rays" % (got["prim"] != ref["prim"]).sum()
    hit = ref["prim"] >= 0
    assert hit.sum() > 1000
    for k in ("t", "u", "v"):  # same arithmetic -> same bits (the sign of a zero included)
        assert np.array_equal(np.ascontiguousarray(got[k][hit]).view(np.uint32), np.ascontiguousarray(ref[k][hit]).view(np.uint32)), k


@pytest.mark.parametrize("which", ["cornell", "materials"])
def test_trace_any_hit_matches_oracle(ctx, oracle_mod, cornell, materials_scene, which):
    sc = cornell if which == "cornell" else materials_scene
    o = oracle_mod.Oracle(sc)
    ctx.upload_scene(sc)
    rays = random_rays(20000, 5)
    rays[:, 3] = 0.01
    rays[:, 7] = np.random.RandomState(1).uniform(0.05, 3.0, len(rays))
    got = ctx.trace(rays, any_hit=True)
    ref = o.trace(rays, any_hit=True)
    assert (got["prim"] == ref["prim"]).all()
    assert 0 < (ref["prim"] == 0).sum() < len(rays)


def test_render_cornell_config1(ctx, oracle_mod, cornell):
    """BASELINE config 1 (Cornell 128x128, 1 spp, timestamp 0) on the GPU vs the oracle."""
    ctx.upload_scene(cornell)
    ctx.frame_begin(128, 128)
    ctx.render(spp=1)
    img = ctx.download().reshape(-1, 4)
    ref, st = oracle_mod.Oracle(cornell).render(128, 128, spp=1)
    e = rmse(img, ref)
    nbad = int((np.abs(img - ref).max(1) > 0).sum())
    print("cornell 1spp rmse %.3e, %d pixels differ" % (e, nbad))
    assert e < TOL_RMSE
    gst = ctx.stats()
    assert gst["extension_rays"] == st["extension_rays"]
    assert gst["shaded_vertices"] == st["shaded_vertices"]


def test_render_progressive_matches_single_call(ctx, oracle_mod, cornell):
    """timestamp protocol (PathTracer.cpp:91-92): 3 calls of 1,2,5 spp == one call of 8 spp == oracle."""
    ctx.upload_scene(cornell)
    ctx.frame_begin(64, 64)
    ctx.render(spp=1, first_timestamp=0)
    ctx.render(spp=2, first_timestamp=1)
    ctx.render(spp=5, first_timestamp=3, timestamps_in_flight=2)
    a = ctx.download().copy()
    ctx.frame_begin(64, 64)
    ctx.render(spp=8, first_timestamp=0)
    b = ctx.download()
    assert np.array_equal(a, b)
    ref, _ = oracle_mod.Oracle(cornell).render(64, 64, spp=8)
    assert rmse(b, ref) < TOL_RMSE


def test_render_materials_full_bsdf_set(ctx, oracle_mod, materials_scene):
    """All eight BSDF types + NEE + deep dielectric paths, 16 spp."""
    ctx.upload_scene(materials_scene)
    ctx.frame_begin(96, 96)
    ctx.render(spp=16)
    img = ctx.download().reshape(-1, 4)
    ref, st = oracle_mod.Oracle(materials_scene).render(96, 96, spp=16)
    e = rmse(img, ref)
    nbad = int((np.abs(img - ref).max(1) > 0).sum())
    print("materials 16spp rmse %.3e, %d pixels differ" % (e, nbad))
    assert e < TOL_RMSE
    assert np.isfinite(img).all()


def test_pixel_subset_reproduces_full_frame(ctx, cornell):
    """Any partition of the frame gives the same pixels (seed = f(pixel, timestamp), raygen.rgen:37)."""
    from gpuspectral_amd import scenes

    W, H = 96, 64
    ctx.upload_scene(cornell)
    ctx.frame_begin(W, H)
    ctx.render(spp=3)
    full = ctx.download().reshape(-1, 4)
    out = np.zeros_like(full)
    for rank in range(3):
        ids = scenes.tile_pixel_ids(W, H, rank, 3, tile=16)
        ctx.frame_begin(W, H, ids)
        ctx.render(spp=3)
        out[ids] = ctx.download_compact()
    assert np.array_equal(out, full)


def test_edge_cases(ctx, oracle_mod):
    """Empty scene, no lights, single triangle, max depth 0."""
    from gpuspectral_amd import abi, scenes

    empty = abi.SceneArrays()
    ctx.upload_scene(empty)
    ctx.frame_begin(16, 16)
    ctx.render(spp=2)
    img = ctx.download()
    assert (img[..., :3] == 0).all()
    assert (ctx.trace(random_rays(100, 1))["prim"] == -1).all()

    b = scenes.SceneBuilder()
    m = b.add_mesh(*scenes.rect_mesh())
    b.add_object(m, scenes.trs((0, 1, 0)), b.diffuse((0.5, 0.5, 0.5)))  # no lights: numLights == 0
    b.to_world = scenes.rowmajor(scenes._CORNELL["camera"])
    sc = b.build()
    ctx.upload_scene(sc)
    ctx.frame_begin(32, 32)
    ctx.render(spp=2)
    img = ctx.download().reshape(-1, 4)
    ref, _ = oracle_mod.Oracle(sc).render(32, 32, spp=2)
    assert np.array_equal(img, ref)

    cm = scenes.cornell_materials(8)
    ctx.upload_scene(cm)
    ctx.frame_begin(48, 48)
    ctx.render(spp=2, max_depth=0, rr_start_depth=0)
    p = abi.default_render_params()
    p.max_depth, p.rr_start_depth = 0, 0
    ref, _ = oracle_mod.Oracle(cm).render(48, 48, spp=2, params=p)
    assert rmse(ctx.download(), ref) < TOL_RMSE


def test_cpp_host_pathtracer_frame_protocol(oracle_mod, cornell):
    """C++ host layer end to end: loadScene(XML) -> PathTracer::createRenderPass x N == oracle,
    with the reference's per-frame timestamp protocol (PathTracer.cpp:91-92)."""
    from conftest import CORNELL_XML
    from gpuspectral_amd import host

    scene = host.Scene(CORNELL_XML)
    pt = host.PathTracer(64, 64)
    for i in range(3):
        assert pt.timestamp == i
        pt.create_render_pass(scene)  # one traceRays(W,H) = +1 spp
    pt.render(scene, 5)
    assert pt.timestamp == 8
    img = pt.download().reshape(-1, 4)
    ref, _ = oracle_mod.Oracle(cornell).render(64, 64, spp=8)
    assert rmse(img, ref) < TOL_RMSE
    assert np.array_equal(img, ref)
    st = pt.stats()
    assert st["num_triangles"] == 36 and st["samples"] == 64 * 64 * 8
    pt.reset()
    assert pt.timestamp == 0 and (pt.download() == 0).all()
    pt.close()


def test_size_independent_properties_at_scale(ctx):
    """Properties that need no oracle, at a size the oracle would take minutes for:
    determinism, split-invariance of the running mean, energy bounds, stats consistency."""
    from gpuspectral_amd import scenes

    sc = scenes.interior(120_000)
    ctx.upload_scene(sc)
    ctx.reset_stats()
    ctx.frame_begin(480, 270)
    ctx.render(spp=6)
    a = ctx.download().copy()
    st = ctx.stats()
    ctx.frame_begin(480, 270)
    ctx.render(spp=2, timestamps_in_flight=1)
    ctx.render(spp=4, first_timestamp=2, timestamps_in_flight=3)
    b = ctx.download()
    assert np.array_equal(a, b)  # same image whatever the batching
    assert np.isfinite(a).all() and (a[..., :3] >= 0).all() and (a[..., 3] == 1).all()
    assert a[..., :3].max() < 20.0 * 52  # firefly clamp bounds every bounce's contribution
    assert st["samples"] == 480 * 270 * 6
    assert st["extension_rays"] >= st["samples"] and st["shaded_vertices"] <= st["extension_rays"]
    assert st["shadow_rays"] <= st["shaded_vertices"]
    assert st["num_triangles"] == sc.num_triangles
    assert sc.num_triangles / 8 < st["num_bvh_nodes"] < sc.num_triangles  # wide nodes: 2..4 children each


def test_render_interior_parity_with_oracle(ctx, oracle_mod):
    """Larger scene (tens of thousands of triangles, 7 material kinds, deep LBVH): image vs oracle.
    Guards the conservativeness of the GPU box tests (an over-eager cull would change a hit)."""
    from gpuspectral_amd import scenes

    sc = scenes.interior(40_000)
    ctx.upload_scene(sc)
    ctx.frame_begin(160, 90)
    ctx.render(spp=3)
    img = ctx.download().reshape(-1, 4)
    o = oracle_mod.Oracle(sc)
    ref, _ = o.render(160, 90, spp=3)
    e = rmse(img, ref)
    nbad = int((np.abs(img - ref).max(1) > 0).sum())
    print("interior 3spp rmse %.3e, %d pixels differ" % (e, nbad))
    assert e < TOL_RMSE
    rays = random_rays(200000, 17, lo=(-4, 0, -5), hi=(4, 2.6, 5))
    got, want = ctx.trace(rays), o.trace(rays)
    assert (got["prim"] == want["prim"]).all()
    hit = want["prim"] >= 0
    assert (got["t"][hit] == want["t"][hit]).all()


@pytest.mark.parametrize("name,tris,lights", [("living-room_shapes", 296416, 512), ("staircase2_shapes", 31247, 336)])
def test_reference_scenes_with_builtin_shapes(ctx, oracle_mod, name, tris, lights):
    """SURVEY 8(f).1 finished: the two shipped scenes whose emitters sit on `sphere` / `disk` shapes,
    loaded with LoadOptions::builtinShapes (tests/golden/ref_scenes/*_shapes.npz = the C++ loader's output, pinned against
    the numpy restatement in tests/test_abi_and_host.py).  'The Modern Living Room' is LIT now (its frame was black), 'Modern
    Hall' has its five ceiling disks: frames and ray counts equal the oracle's bit for bit."""
    from conftest import GOLDEN
    from gpuspectral_amd import abi

    sc = abi.SceneArrays.load(os.path.join(GOLDEN, "ref_scenes", name + ".npz"))
    assert sc.num_triangles == tris and len(sc.lights) == lights
    ctx.upload_scene(sc)
    o = oracle_mod.Oracle(sc)
    W, H, spp = 256, 144, 3
    ctx.frame_begin(W, H)
    ctx.reset_stats()
    ctx.render(spp=spp)
    img = ctx.download().reshape(-1, 4)
    st = ctx.stats()
    ref, ost = o.render(W, H, spp=spp)
    assert np.array_equal(img, ref)
    assert st["extension_rays"] == ost["extension_rays"] and st["shadow_rays"] == ost["shadow_rays"] > W * H
    lit = (img[:, :3] > 0).any(1).mean()
    assert lit > 0.5, "only %.0f %% of the pixels received light" % (100 * lit)


def test_living_room_reference_scene(ctx, oracle_mod):
    """SURVEY 8(f).1, the third of the reference's large shipped scenes: 'The Modern Living Room' as the product's C++
    loader flattens it (tests/golden/ref_scenes/living-room.npz: 295 904 triangles, 28 instances; diffuse, dielectric,
    rough conductor and rough plastic records; the seven meshes listed in the reference tree's .MISSING_LARGE_BLOBS are
    absent there and therefore here).  What the reference can show of this scene is dark: its ONLY emitter is an area
    light on a <shape type="sphere">, a shape kind the reference's loader skips (S/engine/Loader.cpp:276-283), so the
    light list is empty, no instance emits, and every sample is black -- a comparison with the Tungsten render shipped
    beside the scene would compare Tungsten's sphere light with nothing.  What the scene does exercise is the geometry
    path on real, artist-made meshes: 300 000 rays against the oracle (closest hit: triangle and t bit for bit; any
    hit), the camera's view of it (the primary-ray hit map through a 1-spp render's ray counts), and the empty light
    list (sampleLight must not divide by the light count)."""
    from conftest import GOLDEN
    from gpuspectral_amd import abi

    sc = abi.SceneArrays.load(os.path.join(GOLDEN, "ref_scenes", "living-room.npz"))
    assert sc.num_triangles == 295904 and len(sc.instances) == 28 and len(sc.lights) == 0
    assert [len(b) for b in sc.bsdfs] == [5, 9, 0, 0, 2, 0, 0, 12]
    ctx.upload_scene(sc)
    o = oracle_mod.Oracle(sc)
    lo, hi = sc.positions.min(0), sc.positions.max(0)  # (object space; every instance of this scene has an identity-like transform)
    rays = random_rays(300000, 23, lo=tuple(lo - 0.1), hi=tuple(hi + 0.1))
    got, want = ctx.trace(rays), o.trace(rays)
    assert (got["prim"] == want["prim"]).all()
    hit = want["prim"] >= 0
    assert 0.3 < hit.mean() < 1.0
    assert (got["t"][hit] == want["t"][hit]).all() and (got["u"][hit] == want["u"][hit]).all() and (got["v"][hit] == want["v"][hit]).all()
    rays[:, 3], rays[:, 7] = 0.01, 2.0
    assert ((ctx.trace(rays, any_hit=True)["prim"] >= 0) == (o.trace(rays, any_hit=True)["prim"] >= 0)).all()
    W, H = 320, 180
    ctx.frame_begin(W, H)
    ctx.reset_stats()
    ctx.render(spp=2)
    img = ctx.download().reshape(-1, 4)
    st = ctx.stats()
    ref, ost = o.render(W, H, spp=2)
    assert np.array_equal(img, ref) and not img[:, :3].any()  # no emitter the loader can load: black, as the reference's
    assert st["extension_rays"] == ost["extension_rays"] > 2 * W * H and st["shadow_rays"] == ost["shadow_rays"] == 0
    # The path rays themselves (the image is black, so only the ray counts above would notice a wrong hit): every
    # extension ray the oracle traced for this frame, in its order and in a shuffled one.  Until r03 ONE of these
    # 918 614 rays had an order-dependent answer -- it meets a 6.3 m x 2 mm bevel whose float32 t comes out 5e-4 short,
    # outside the triangle's padded box, so the bevel won or lost against the wall behind it depending on which was
    # tested first (r02's kernel and the oracle disagreed); the boxes of slivers are padded by their aspect ratio now
    # (pt_bvh.hip k_bake, oracle buildAccel) and the smaller float t wins in every order.
    rays = o.extension_rays_of(W, H, spp=2)
    assert len(rays) == ost["extension_rays"]
    want = o.trace(rays)
    got = ctx.trace(rays)
    assert (got["prim"] == want["prim"]).all() and (got["t"] == want["t"])[want["prim"] >= 0].all()
    perm = np.random.RandomState(5).permutation(len(rays))
    again = ctx.trace(rays[perm])
    assert (again["prim"] == got["prim"][perm]).all() and (again["t"] == got["t"][perm]).all()
    sliver = np.array([[-0.8807780146598816, 0.00010001489135902375, -0.18816836178302765, 0.0,
                        -0.6416327953338623, 0.45284968614578247, 0.6190593838691711, 1e10]], np.float32)
    assert ctx.trace(sliver)["prim"][0] == o.trace(sliver)["prim"][0] == 295778


@pytest.mark.parametrize("name", ["coffee", "staircase2", "interior", "caustics"])
def test_path_rays_hit_the_same_triangle_in_any_order(ctx, oracle_mod, name):
    """The rays paths actually shoot (the oracle's extension rays of a small frame: they start ON surfaces and graze
    their neighbours, unlike random rays in a box), traced by the GPU in the oracle's order and in a shuffled one:
    triangle and t must agree bit for bit, i.e. the closest hit is a function of the ray, not of the traversal."""
    from conftest import GOLDEN
    from gpuspectral_amd import abi, scenes

    if name in ("coffee", "staircase2"):
        sc = abi.SceneArrays.load(os.path.join(GOLDEN, "ref_scenes", name + ".npz"))
    else:
        sc = scenes.interior(150_000, seed=7) if name == "interior" else scenes.caustics(150_000, seed=11)
    ctx.upload_scene(sc)
    o = oracle_mod.Oracle(sc)
    rays = o.extension_rays_of(200, 120, spp=2)
    assert len(rays) > 2 * 200 * 120
    want, got = o.trace(rays), ctx.trace(rays)
    assert (got["prim"] == want["prim"]).all() and (got["t"] == want["t"])[want["prim"] >= 0].all()
    perm = np.random.RandomState(9).permutation(len(rays))
    again = ctx.trace(rays[perm])
    assert (again["prim"] == got["prim"][perm]).all() and (again["t"] == got["t"][perm]).all()


@pytest.mark.parametrize("size", [(1, 1), (3, 2), (17, 5), (40, 23), (65, 31)])
def test_tiny_frames_and_small_queues(ctx, oracle_mod, materials_scene, size):
    """Queues of a handful to a few thousand rays: every hand-out shard of the persistent traversal
    kernel must be served whatever the launch size (a missed chunk leaves stale hit records)."""
    W, H = size
    ctx.upload_scene(materials_scene)
    ctx.frame_begin(W, H)
    ctx.render(spp=5)
    img = ctx.download().reshape(-1, 4)
    ref, st = oracle_mod.Oracle(materials_scene).render(W, H, spp=5)
    assert np.array_equal(img, ref)
    ctx.frame_begin(W, H)
    ctx.reset_stats()
    ctx.render(spp=5)
    assert np.array_equal(ctx.download().reshape(-1, 4), ref)
    assert ctx.stats()["extension_rays"] == st["extension_rays"]


def test_deep_dielectric_paths_max_depth_32(ctx, oracle_mod):
    """BASELINE config 5 at test size: dielectric-heavy scene, max_depth 32 (all-delta vertices skip NEE,
    Russian roulette from depth 11, depth cap)."""
    from gpuspectral_amd import abi, scenes

    sc = scenes.caustics(6000)
    ctx.upload_scene(sc)
    ctx.frame_begin(96, 96)
    ctx.render(spp=4, max_depth=32)
    p = abi.default_render_params()
    p.max_depth = 32
    ref, st = oracle_mod.Oracle(sc).render(96, 96, spp=4, params=p)
    img = ctx.download().reshape(-1, 4)
    assert rmse(img, ref) < TOL_RMSE
    assert np.array_equal(img, ref)
    assert ctx.stats()["shadow_rays"] > 0


def test_checkpoint_resume(oracle_mod, cornell):
    """Accumulate buffer + next timestamp are the whole integrator state (checkpoint/resume):
    a second context resumed from a downloaded buffer continues bit-exactly."""
    import gpuspectral_amd as g

    a = g.Context(0)
    a.upload_scene(cornell)
    a.frame_begin(48, 48)
    a.render(spp=3)
    saved = a.download_compact().copy()
    a.close()
    b = g.Context(0)
    b.upload_scene(cornell)
    b.frame_begin(48, 48)
    b.upload_accum(saved)
    b.render(spp=4, first_timestamp=3)
    img = b.download().reshape(-1, 4)
    b.close()
    ref, _ = oracle_mod.Oracle(cornell).render(48, 48, spp=7)
    assert np.array_equal(img, ref)


def test_pipelined_calls_without_sync(ctx, oracle_mod, materials_scene):
    """gsp_render only queues work: paths of earlier calls stay in flight while later calls inject theirs
    (the reference's one-createRenderPass-per-frame loop, main.cpp:20-28).  24 one-sample calls, a change of
    max_depth in between (forces a drain), and the ray counters must equal the all-at-once result."""
    W, H = 72, 40
    ctx.upload_scene(materials_scene)
    ctx.frame_begin(W, H)
    ctx.reset_stats()
    for t in range(24):
        ctx.render(spp=1, first_timestamp=t)
    st_a = ctx.stats()
    a = ctx.download().copy()
    ctx.frame_begin(W, H)
    ctx.reset_stats()
    ctx.render(spp=24)
    st_b = ctx.stats()
    b = ctx.download().copy()
    assert np.array_equal(a, b)
    for k in ("extension_rays", "shadow_rays", "samples"):
        assert st_a[k] == st_b[k], k
    ref, st_o = oracle_mod.Oracle(materials_scene).render(W, H, spp=24)
    assert np.array_equal(b.reshape(-1, 4), ref)
    assert st_b["extension_rays"] == st_o["extension_rays"] and st_b["shadow_rays"] == st_o["shadow_rays"]
    # integrator constants change while paths are in flight: earlier samples keep the old constants
    ctx.frame_begin(W, H)
    ctx.render(spp=5, first_timestamp=0, max_depth=3)
    ctx.render(spp=5, first_timestamp=5, max_depth=50)
    c = ctx.download().copy()
    from gpuspectral_amd import abi

    o = oracle_mod.Oracle(materials_scene)
    p = abi.default_render_params(5, 0)
    p.max_depth = 3
    acc, _ = o.render(W, H, spp=5, params=p)
    p.max_depth = 50
    acc, _ = o.render(W, H, spp=5, first_timestamp=5, params=p, accum=acc)
    assert np.array_equal(c.reshape(-1, 4), acc)


def _soup_scene():
    """Triangle soup built to stress the claim that hits do not depend on BVH topology (the oracle walks a
    binned-SAH BVH2, the GPU a compressed PLOC BVH4): degenerate and duplicated triangles, shared edges, extreme
    scales, mirrored / non-uniformly scaled instances."""
    from gpuspectral_amd import scenes

    rng = np.random.RandomState(11)
    b = scenes.SceneBuilder()
    mat = b.diffuse((0.5, 0.5, 0.5))

    def add(tris, transform=None):
        tris = np.asarray(tris, np.float32).reshape(-1, 3)
        n = np.zeros_like(tris)
        n[:, 1] = 1.0
        b.add_object(b.add_mesh(tris, n), scenes.trs() if transform is None else transform, mat)

    c = rng.uniform(-1, 1, (3000, 1, 3))
    soup = c + rng.normal(size=(3000, 3, 3)) * rng.choice([0.02, 0.1, 0.5], (3000, 1, 1))
    add(soup)
    add(soup[:100])  # exact duplicates in a second instance: equal t, the smaller global triangle id must win
    deg = soup[100:300].copy()
    deg[:100, 2] = deg[:100, 1]  # two equal vertices
    deg[100:, 2] = deg[100:, 0] + 2.0 * (deg[100:, 1] - deg[100:, 0])  # collinear
    add(deg)
    # a finely tessellated sheet: rays through shared edges / vertices must hit exactly one or the other, never slip through
    g = np.linspace(-1, 1, 33, dtype=np.float32)
    quads = []
    for i in range(32):
        for j in range(32):
            p00, p10, p01, p11 = (g[i], 0.25, g[j]), (g[i + 1], 0.25, g[j]), (g[i], 0.25, g[j + 1]), (g[i + 1], 0.25, g[j + 1])
            quads += [p00, p10, p11, p00, p11, p01]
    add(quads)
    add(soup[300:800] * 1e-4, scenes.trs((100.0, 100.0, 100.0)))  # tiny geometry far from the origin
    add(soup[800:1000], scenes.trs((0, 0, 0), (1e4, 1e4, 1e4)))  # huge triangles
    add(soup[1000:1500], scenes.trs((0.3, -0.2, 0.1), (-1.0, 2.5, 0.4), 37.0))  # mirrored, non-uniform scale, rotated
    return b.build()


def _soup_rays():
    rng = np.random.RandomState(12)
    rays = [random_rays(30000, 21, lo=(-2, -2, -2), hi=(2, 2, 2))]
    ax = random_rays(6000, 22, lo=(-1, -1, -1), hi=(1, 1, 1))  # axis-parallel: zero direction components, 1/d = inf
    for i in range(len(ax)):
        d = np.zeros(3, np.float32)
        d[i % 3] = 1.0 if (i // 3) % 2 else -1.0
        ax[i, 4:7] = d
    rays.append(ax)
    gr = np.zeros((4000, 8), np.float32)  # straight down through grid vertices and edges of the sheet
    gv = np.linspace(-1, 1, 33, dtype=np.float32)
    gr[:, 0] = gv[rng.randint(0, 33, 4000)]
    gr[:, 2] = np.where(rng.rand(4000) < 0.5, gv[rng.randint(0, 33, 4000)], rng.uniform(-1, 1, 4000)).astype(np.float32)
    gr[:, 1] = 3.0
    gr[:, 5] = -1.0
    gr[:, 7] = 1e10
    rays.append(gr)
    far = random_rays(4000, 23, lo=(99.9999, 99.9999, 99.9999), hi=(100.0001, 100.0001, 100.0001))
    rays.append(far)
    out = random_rays(4000, 24, lo=(-3e4, -3e4, -3e4), hi=(3e4, 3e4, 3e4))
    out[:, 4:7] = -out[:, 0:3] / np.linalg.norm(out[:, 0:3], axis=1, keepdims=True)  # from far outside towards the origin
    rays.append(out)
    return np.concatenate(rays).astype(np.float32)


def test_triangle_soup_hits_do_not_depend_on_bvh(ctx, oracle_mod):
    sc = _soup_scene()
    o = oracle_mod.Oracle(sc)
    ctx.upload_scene(sc)
    rays = _soup_rays()
    got, ref = ctx.trace(rays), o.trace(rays)
    bad = got["prim"] != ref["prim"]
    assert not bad.any(), "closest hit differs on %d of %d rays, first %s" % (bad.sum(), len(rays), np.nonzero(bad)[0][:5])
    hit = ref["prim"] >= 0
    assert hit.sum() > 10000
    for k in ("t", "u", "v"):  # bitwise: rays through shared vertices / edges report u or v = 0 with the oracle's sign
        assert np.array_equal(np.ascontiguousarray(got[k][hit]).view(np.uint32), np.ascontiguousarray(ref[k][hit]).view(np.uint32)), k
    # the sheet is watertight: every vertical grid ray inside it is stopped at or above y = 0.25
    n0 = 36000
    sheet = ref[n0 : n0 + 4000]
    assert (sheet["prim"] >= 0).all() and (sheet["t"] <= np.float32(2.75)).all()
    sh = rays.copy()
    sh[:, 3] = 0.01
    sh[:, 7] = np.random.RandomState(3).uniform(0.05, 5.0, len(sh)).astype(np.float32)
    g2, r2 = ctx.trace(sh, any_hit=True), o.trace(sh, any_hit=True)
    assert (g2["prim"] == r2["prim"]).all()


@pytest.mark.parametrize("ntris", [1, 3, 40, 100, 126, 128, 130, 190, 260, 700])
def test_trees_around_the_size_of_the_lds_node_copy(ctx, oracle_mod, ntris):
    """k_trace keeps the first 64 records of the node array in LDS and fetches the others from memory (pt_wavetrace.h).
    Trees with fewer records than that, exactly about that many and a few times more: closest hits bit for bit, any-hit
    verdicts, and a small frame."""
    from gpuspectral_amd import scenes

    rng = np.random.RandomState(1000 + ntris)
    b = scenes.SceneBuilder()
    c = rng.uniform(-1, 1, (ntris, 1, 3))
    tris = (c + rng.normal(size=(ntris, 3, 3)) * 0.15).astype(np.float32).reshape(-1, 3)
    nrm = np.zeros_like(tris)
    nrm[:, 1] = 1.0
    b.add_object(b.add_mesh(tris, nrm), scenes.trs(), b.diffuse((0.6, 0.6, 0.6)))
    b.add_object(b.add_mesh(*scenes.rect_mesh()), scenes.trs((0, 1.8, 0), 0.5, 0.0), b.diffuse((0, 0, 0)), twofaced=True, emission=(9, 9, 9))
    b.camera_lookat((0.2, 0.4, 3.5), (0, 0, 0), fov_deg=45)
    sc = b.build()
    o = oracle_mod.Oracle(sc)
    ctx.upload_scene(sc)
    st = ctx.stats()
    assert st["num_triangles"] == ntris + 2
    rays = random_rays(20000, 5 + ntris, lo=(-2, -2, -2), hi=(2, 2, 2))
    got, ref = ctx.trace(rays), o.trace(rays)
    assert (got["prim"] == ref["prim"]).all()
    hit = ref["prim"] >= 0
    for k in ("t", "u", "v"):
        assert np.array_equal(np.ascontiguousarray(got[k][hit]).view(np.uint32), np.ascontiguousarray(ref[k][hit]).view(np.uint32)), k
    sh = rays.copy()
    sh[:, 3], sh[:, 7] = 0.01, 2.5
    assert ((ctx.trace(sh, any_hit=True)["prim"] >= 0) == (o.trace(sh, any_hit=True)["prim"] >= 0)).all()
    ctx.frame_begin(48, 32)
    ctx.render(spp=4)
    ref_img, _ = o.render(48, 32, spp=4)
    assert np.array_equal(ctx.download().reshape(-1, 4), ref_img)


@pytest.mark.parametrize("rounds", ["0", "1", "17"])
def test_reinsertion_rounds_change_the_tree_not_the_hits(ctx, oracle_mod, materials_scene, monkeypatch, rounds):
    """The BVH build runs parallel reinsertion rounds behind PLOC (pt_bvh.hip k_ri_*; GSP_BVH_REINSERT overrides the default 6):
    whatever their number, closest hits, any-hit verdicts and a frame equal the oracle's -- on the full-BSDF scene and on the
    triangle soup with its duplicates, degenerate triangles and mirrored instances."""
    monkeypatch.setenv("GSP_BVH_REINSERT", rounds)
    for sc, rays in ((materials_scene, random_rays(30000, 31)), (_soup_scene(), _soup_rays())):
        o = oracle_mod.Oracle(sc)
        ctx.upload_scene(sc)
        got, ref = ctx.trace(rays), o.trace(rays)
        assert (got["prim"] == ref["prim"]).all()
        hit = ref["prim"] >= 0
        for k in ("t", "u", "v"):
            assert np.array_equal(np.ascontiguousarray(got[k][hit]).view(np.uint32), np.ascontiguousarray(ref[k][hit]).view(np.uint32)), k
        sh = rays.copy()
        sh[:, 3] = 0.01
        sh[:, 7] = np.random.RandomState(4).uniform(0.05, 4.0, len(sh)).astype(np.float32)
        assert (ctx.trace(sh, any_hit=True)["prim"] == o.trace(sh, any_hit=True)["prim"]).all()
    ctx.upload_scene(materials_scene)
    ctx.frame_begin(64, 48)
    ctx.render(spp=4)
    ref_img, _ = oracle_mod.Oracle(materials_scene).render(64, 48, spp=4)
    assert np.array_equal(ctx.download().reshape(-1, 4), ref_img)


def _nested_scene(n=520):
    """Concentric, geometrically growing quads + boxes around one point: equal Morton codes and nested boxes make the
    agglomerative build chain them up, so the BVH is far deeper than the 16 stack levels a lane keeps in LDS."""
    from gpuspectral_amd import scenes

    b = scenes.SceneBuilder()
    rect, box = b.add_mesh(*scenes.rect_mesh()), b.add_mesh(*scenes.box_mesh())
    white, mirror, glass = b.diffuse((0.7, 0.7, 0.7)), b.mirror(0.0), b.dielectric(1.5)
    for k in range(n):
        s = 0.004 * 1.01 ** k  # 4 mm ... 0.7 m
        m = (white, mirror, glass)[k % 3]
        b.add_object(rect if k % 2 else box, scenes.trs((0, 1, 0), s, 17.0 * k), m, twofaced=True)
    b.add_object(box, scenes.trs((0.5, 2.6, 0.8), 0.2, 0.0), b.diffuse((0, 0, 0)), twofaced=False, emission=(15, 14, 12))
    b.add_object(box, scenes.trs((0, 1, 0), (150, 150, 150)), white, twofaced=True)  # the room around it all
    b.camera_lookat((0.3, 1.2, 4.0), (0, 1, 0), fov_deg=50)
    return b.build()


def test_deep_tree_runs_on_the_stack_spill_path(ctx, oracle_mod):
    """A tree deeper than the LDS stack levels of k_trace (20) and than k_finish's stack (36): the per-lane stack continues
    in HBM (slow path of WaveStack) and the drain is left to the wavefront kernels; rays, image and ray counts must still
    equal the oracle's."""
    sc = _nested_scene()
    ctx.upload_scene(sc)
    o = oracle_mod.Oracle(sc)
    rng = np.random.RandomState(41)
    rays = random_rays(150000, 41, lo=(-3, -2, -3), hi=(3, 4, 3))
    aim = np.array([0, 1, 0], np.float32) - rays[:, 0:3] + rng.normal(scale=0.15, size=(len(rays), 3)).astype(np.float32)
    rays[:, 4:7] = aim / np.linalg.norm(aim, axis=1, keepdims=True)  # through the nest, where the tree is deep
    got, want = ctx.trace(rays), o.trace(rays)
    assert (got["prim"] == want["prim"]).all() and (got["t"] == want["t"])[want["prim"] >= 0].all()
    sh = rays.copy()
    sh[:, 3], sh[:, 7] = 0.01, 3.0
    assert ((ctx.trace(sh, any_hit=True)["prim"] >= 0) == (o.trace(sh, any_hit=True)["prim"] >= 0)).all()
    W, H = 96, 64
    ctx.frame_begin(W, H)
    ctx.reset_stats()
    ctx.render(spp=3)
    img = ctx.download().reshape(-1, 4)
    st = ctx.stats()
    assert st["bvh_depth"] > 36, st["bvh_depth"]  # (else this test does not reach the paths it is about)
    ref, ost = o.render(W, H, spp=3)
    assert np.array_equal(img, ref)
    assert st["extension_rays"] == ost["extension_rays"] and st["shadow_rays"] == ost["shadow_rays"]


def test_two_pipeline_lanes_give_the_same_image(oracle_mod, materials_scene, monkeypatch):
    """GSP_LANES=2 deals the owned pixels to two independent pipelines on two streams; the per-pixel arithmetic
    is unchanged, so the frame, the ray counts and a pixel-subset context are bit-identical to one lane."""
    import gpuspectral_amd as g
    from gpuspectral_amd.scenes import tile_pixel_ids

    W, H, spp = 71, 45, 9  # odd pixel count: the lanes own different numbers of pixels
    out = {}
    for lanes in ("1", "2"):
        monkeypatch.setenv("GSP_LANES", lanes)
        with g.Context(0) as c:
            c.upload_scene(materials_scene)
            c.frame_begin(W, H)
            for t in range(0, spp, 3):
                c.render(spp=3, first_timestamp=t)
            st = c.stats()
            out[lanes] = (c.download().copy(), st["extension_rays"], st["shadow_rays"], st["samples"])
            ids = tile_pixel_ids(W, H, 1, 3, tile=8)
            c.frame_begin(W, H, ids)
            c.render(spp=spp)
            sub = c.download_compact()
            assert np.array_equal(sub, out[lanes][0].reshape(-1, 4)[ids])
    assert np.array_equal(out["1"][0], out["2"][0])
    assert out["1"][1:] == out["2"][1:]
    ref, st_o = oracle_mod.Oracle(materials_scene).render(W, H, spp=spp)
    assert np.array_equal(out["2"][0].reshape(-1, 4), ref)
    assert out["2"][1] == st_o["extension_rays"] and out["2"][2] == st_o["shadow_rays"]


def test_cpp_cli_renders_the_reference_scene(tmp_path):
    """`gsp_render scene.xml out.pfm W H spp` (the replacement of the reference's main.cpp loop): loader -> C ABI ->
    PFM writer, compared with the committed Config-1 fixture."""
    import os
    import subprocess

    from conftest import CORNELL_XML, ROOT

    lib = os.path.join(ROOT, "gpuspectral_amd", "lib")
    out = str(tmp_path / "cornell.pfm")
    env = dict(os.environ, LD_LIBRARY_PATH=lib + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([os.path.join(lib, "gsp_render"), CORNELL_XML, out, "128", "128", "1"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    with open(out, "rb") as f:
        assert f.readline() == b"PF\n"
        w, h = map(int, f.readline().split())
        assert float(f.readline()) < 0  # little endian
        img = np.frombuffer(f.read(), np.float32).reshape(h, w, 3)[::-1]  # PFM rows run bottom-up
    ref = np.load(os.path.join(ROOT, "tests", "golden", "cornell_128_1spp.npy"))
    assert np.array_equal(img.reshape(-1, 3), ref.reshape(-1, ref.shape[-1])[:, :3])


def test_cpp_cli_options(tmp_path, oracle_mod, cornell):
    """`gsp_render --no-nee --memory-share F --pool-paths N`: the C++ host passes gsp_ctx_options and RenderParams.nee through
    (r04); the frame equals the oracle's nee = 0 frame whatever the pool size; an unknown option is a usage error."""
    import os
    import subprocess

    from conftest import CORNELL_XML, ROOT
    from gpuspectral_amd import abi

    lib = os.path.join(ROOT, "gpuspectral_amd", "lib")
    out = str(tmp_path / "cornell_nonee.pfm")
    env = dict(os.environ, LD_LIBRARY_PATH=lib + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([os.path.join(lib, "gsp_render"), "--no-nee", "--memory-share", "0.05", "--pool-paths", "100000", CORNELL_XML, out,
                        "96", "64", "3"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    with open(out, "rb") as f:
        assert f.readline() == b"PF\n"
        w, h = map(int, f.readline().split())
        f.readline()
        img = np.frombuffer(f.read(), np.float32).reshape(h, w, 3)[::-1]
    p = abi.default_render_params()
    p.disable_nee = 1
    ref, _ = oracle_mod.Oracle(cornell).render(96, 64, spp=3, params=p)
    assert np.array_equal(img.reshape(-1, 3), ref[:, :3])
    r = subprocess.run([os.path.join(lib, "gsp_render"), "--frobnicate", CORNELL_XML, out], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "unknown option" in r.stderr


def test_cpp_cli_builtin_shapes(tmp_path, oracle_mod, staircase2_xml):
    """`gsp_render --builtin-shapes` (LoadOptions::builtinShapes, r05): the C++ host end to end on 'Modern Hall' with its five
    `disk` ceiling lights built -- the frame equals the oracle on the numpy loader's arrays of the same options, and differs
    from the default load (the reference's: disks skipped)."""
    import os
    import subprocess

    from conftest import ROOT
    from oracle import mitsuba_loader as ml

    lib = os.path.join(ROOT, "gpuspectral_amd", "lib")
    env = dict(os.environ, LD_LIBRARY_PATH=lib + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    imgs = {}
    for name, flags in (("shapes", ["--builtin-shapes"]), ("plain", [])):
        out = str(tmp_path / (name + ".pfm"))
        r = subprocess.run([os.path.join(lib, "gsp_render")] + flags + [staircase2_xml, out, "80", "60", "2"], env=env, capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        with open(out, "rb") as f:
            assert f.readline() == b"PF\n"
            w, h = map(int, f.readline().split())
            f.readline()
            imgs[name] = np.frombuffer(f.read(), np.float32).reshape(h, w, 3)[::-1].reshape(-1, 3)
    ref, _ = oracle_mod.Oracle(ml.load_scene(staircase2_xml, builtin_shapes=True)).render(80, 60, spp=2)
    assert np.array_equal(imgs["shapes"], ref[:, :3])
    assert not np.array_equal(imgs["shapes"], imgs["plain"])


def test_zeroed_render_params_are_the_reference(ctx, oracle_mod, cornell):
    """a C host that zero-initialises gsp_render_params and fills the fields it knows must get the reference as
    shipped (`#define NEE true`): the appended field is `disable_nee`, 0 = on."""
    from gpuspectral_amd import abi

    p = abi.RenderParams()  # all zero
    p.spp, p.max_depth, p.rr_start_depth, p.clamp = 2, 50, 10, 20.0
    assert p.disable_nee == 0
    ctx.upload_scene(cornell)
    ctx.frame_begin(64, 48)
    ctx.reset_stats()
    ctx.render(spp=2, params=p)
    img = ctx.download().reshape(-1, 4)
    st = ctx.stats()
    ref, ost = oracle_mod.Oracle(cornell).render(64, 48, spp=2)
    assert np.array_equal(img, ref) and st["shadow_rays"] == ost["shadow_rays"] > 0


def test_render_params_struct_size(ctx, oracle_mod, cornell):
    """ABI 8: gsp_render_params is sized by its first field.  A host whose struct is LONGER than the library's (a newer header) is
    read up to the library's size; one shorter than the ABI-8 layout is refused with a message, not read past its end."""
    import ctypes as C

    import gpuspectral_amd as g
    from gpuspectral_amd import abi

    ctx.upload_scene(cornell)
    ctx.frame_begin(32, 24)
    ref, _ = oracle_mod.Oracle(cornell).render(32, 24, spp=2)

    class Longer(C.Structure):
        _fields_ = [("p", abi.RenderParams), ("future_field", C.c_uint32 * 4)]

    big = Longer()
    big.p = abi.default_render_params(2, 0)
    big.p.struct_size = C.sizeof(Longer)
    big.future_field[0] = 0xDEADBEEF  # (not read: beyond the library's struct)
    ctx._check(ctx._L.gsp_render(ctx._h, C.cast(C.byref(big), C.POINTER(abi.RenderParams))), "gsp_render")
    assert np.array_equal(ctx.download().reshape(-1, 4), ref)
    small = abi.default_render_params(2, 0)
    small.struct_size = 12
    with pytest.raises(g.pt.GspError, match="struct_size"):
        ctx.render(spp=2, params=small)


def test_random_scene_fuzz_matches_oracle(ctx, oracle_mod):
    """tests/tools/fuzz_parity.py: random small scenes with all eight BSDF types at ordinary and extreme parameters,
    mirrored / non-uniformly scaled instances, several lights, random cameras: frames (NaN pixels included) and
    ray counts equal the oracle's.  (3 300 seeds were run once by hand: 0 mismatches.)"""
    import os
    import sys

    from conftest import ROOT

    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import fuzz_parity

    for seed in range(5000, 5030):
        ok, ndiff, tris = fuzz_parity.check(ctx, oracle_mod, seed)
        assert ok, "seed %d: %d pixels differ (%d triangles)" % (seed, ndiff, tris)


def test_peek_shows_a_prefix_of_the_timestamps(oracle_mod, materials_scene):
    """gsp_peek: the accumulate buffer without waiting for paths in flight = the running mean of the first k
    timestamps of every pixel, for the k it reports (what the reference's blit pass would show that frame)."""
    import gpuspectral_amd as g

    W, H = 96, 64
    o = oracle_mod.Oracle(materials_scene)
    refs = {}
    with g.Context(0) as c:
        c.upload_scene(materials_scene)
        c.frame_begin(W, H)
        seen = []
        for t in range(12):
            c.render(spp=1, first_timestamp=t, timestamps_in_flight=1)
            img, k = c.peek()
            assert 0 <= k <= t + 1
            seen.append(k)
            if k not in refs:
                refs[k] = o.render(W, H, spp=k)[0] if k else np.zeros((W * H, 4), np.float32)
            assert np.array_equal(img, refs[k]), "peek after %d calls reports %d timestamps" % (t + 1, k)
        assert seen == sorted(seen)
        # (r05) the same view into DEVICE memory: gsp_peek_to_device (no wait for the paths in flight, no trip through the host)
        import ctypes as C

        hip = C.CDLL("libamdhip64.so")
        dptr = C.c_void_p()
        nbytes = W * H * 16
        assert hip.hipMalloc(C.byref(dptr), C.c_size_t(nbytes)) == 0
        try:
            c.render(spp=1, first_timestamp=12, timestamps_in_flight=1)
            kd = c.peek_to_device(dptr.value, nbytes)
            back = np.zeros((W * H, 4), np.float32)
            assert hip.hipMemcpy(C.c_void_p(back.ctypes.data), dptr, C.c_size_t(nbytes), 2) == 0  # hipMemcpyDeviceToHost
            img, k = c.peek()
            assert k >= kd and 0 <= kd <= 13
            if kd not in refs:
                refs[kd] = o.render(W, H, spp=kd)[0]
            assert np.array_equal(back, refs[kd])
            with pytest.raises(g.GspError, match="destination too small"):
                c.peek_to_device(dptr.value, nbytes - 16)
        finally:
            hip.hipFree(dptr)
        full = c.download().reshape(-1, 4)
        img, k = c.peek()
        assert k == 13 and np.array_equal(img, full)
        assert np.array_equal(full, o.render(W, H, spp=13)[0])


def test_tiny_frames_with_many_samples(ctx, oracle_mod):
    """Batches of thousands of timestamps per slot (1x1 x 20 000 spp, 8x8 x 3 000 spp): the running mean over a long
    timestamp range stays bit-identical to the oracle's."""
    from gpuspectral_amd import scenes

    sc = scenes.cornell_materials(8)
    o = oracle_mod.Oracle(sc)
    ctx.upload_scene(sc)
    for W, H, spp in ((1, 1, 20000), (8, 8, 3000)):
        ctx.frame_begin(W, H)
        ctx.render(spp=spp)
        ref, _ = o.render(W, H, spp=spp)
        assert np.array_equal(ctx.download().reshape(-1, 4), ref), (W, H, spp)


def test_many_lights_tables_beyond_the_lds_budget(ctx, oracle_mod):
    """k_shade stages the BSDF + light tables into LDS when they fit 8 KB; an emissive tessellated sphere makes 720
    triangle lights (46 KB), so this scene runs the global-memory path of the same code."""
    from gpuspectral_amd import scenes

    b = scenes.SceneBuilder()
    room = b.add_mesh(*scenes.box_mesh())
    ball = b.add_mesh(*scenes.sphere_mesh(24, 16))
    b.add_object(room, scenes.trs((0, 1, 0), (2.0, 1.2, 2.0)), b.diffuse((0.7, 0.6, 0.5)), twofaced=True)
    b.add_object(ball, scenes.trs((0.3, 1.4, -0.2), (0.3, 0.3, 0.3)), b.diffuse((0, 0, 0)), emission=(60.0, 50.0, 30.0))
    b.add_object(ball, scenes.trs((-0.6, 0.4, 0.5), (0.4, 0.4, 0.4)), b.rough_conductor((0.2, 0.9, 1.1), (3.9, 2.4, 2.2), 0.2, (1, 1, 1)))
    b.camera_lookat((1.6, 1.1, 1.7), (0, 0.9, 0), fov_deg=60)
    sc = b.build()
    assert len(sc.lights) * 64 > 8192
    W, H, spp = 64, 48, 4
    ctx.upload_scene(sc)
    ctx.frame_begin(W, H)
    ctx.reset_stats()
    ctx.render(spp=spp)
    img = ctx.download().reshape(-1, 4)
    st = ctx.stats()
    ref, so = oracle_mod.Oracle(sc).render(W, H, spp=spp)
    assert rmse(img, ref) < TOL_RMSE
    assert np.array_equal(img, ref)
    assert st["shadow_rays"] == so["shadow_rays"] and st["extension_rays"] == so["extension_rays"] and st["shadow_rays"] > 0


def test_random_call_sequences_equal_one_shot_renders(oracle_mod):
    """The streaming pool and the host-one-iteration-behind loop under arbitrary call patterns: random frame sizes, random
    splits of the samples over gsp_render calls with random batch sizes, peeks / syncs / stats reads / downloads in between,
    two contexts interleaved -- the final frame and the ray counts must equal ONE gsp_render call of all samples on a fresh
    context (which the other tests tie to the oracle; the first sequence is also checked against the oracle here)."""
    import sys

    import gpuspectral_amd as g
    from conftest import ROOT

    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import fuzz_parity

    rng = np.random.RandomState(77)
    a, b = g.Context(0), g.Context(0)
    try:
        for seq in range(16):
            sc = fuzz_parity.random_scene(9000 + seq)
            W, H = int(rng.randint(1, 180)), int(rng.randint(1, 120))
            total = int(rng.randint(1, 60))
            tif = int(rng.choice([0, 0, 1, 2, 7]))
            for c in (a, b):
                c.upload_scene(sc)
                c.frame_begin(W, H)
                c.reset_stats()
            b.render(spp=total, timestamps_in_flight=tif)  # one shot
            done = 0
            while done < total:
                k = int(min(total - done, rng.randint(1, 12)))
                a.render(spp=k, first_timestamp=done, timestamps_in_flight=tif)
                done += k
                op = rng.randint(6)
                if op == 0:
                    img, folded = a.peek()
                    assert 0 <= folded <= done
                elif op == 1:
                    a.sync()
                elif op == 2:
                    assert a.stats()["samples"] == W * H * done
                elif op == 3:
                    a.download_compact()
            ia, ib = a.download(), b.download()
            sa, sb = a.stats(), b.stats()
            assert np.array_equal(ia, ib, equal_nan=True), (seq, W, H, total, tif)
            for key in ("extension_rays", "shadow_rays", "shaded_vertices", "samples"):
                assert sa[key] == sb[key], (seq, key)
            if seq == 0:
                ref, _ = oracle_mod.Oracle(sc).render(W, H, spp=total)
                assert np.array_equal(ib.reshape(-1, 4), ref, equal_nan=True)
    finally:
        a.close()
        b.close()


def test_gpu_frames_equal_the_executed_reference_shaders(ctx):
    """The HIP path against frames produced by the reference's OWN shader text -- raygen.rgen / rayhit.rchit / the miss shaders
    compiled as C++ through oracle/glsl_shim.h and run over the oracle's traversal (tests/golden/make_glsl_vectors.py; fixture
    tests/golden/glsl_vectors.npz, the only thing that travels) -- bit for bit, with no oracle render in between on this box."""
    import sys

    from conftest import GOLDEN

    sys.path.insert(0, GOLDEN)
    import make_glsl_vectors as M

    with np.load(os.path.join(GOLDEN, "glsl_vectors.npz")) as z:
        for name, (w, h, spp) in M.IMAGES.items():
            want, rays = z["img_" + name], z["img_" + name + "_rays"]
            ctx.upload_scene(M.image_scene(name))
            ctx.frame_begin(w, h)
            ctx.reset_stats()
            ctx.render(spp=spp)
            got = ctx.download()
            st = ctx.stats()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, int((got != want).any(axis=2).sum()))
            assert (st["extension_rays"], st["shadow_rays"]) == (int(rays[0]), int(rays[1])), name
