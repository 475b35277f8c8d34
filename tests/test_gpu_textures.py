"""SURVEY 8(f).3 -- the reference's dormant features (textures, environment map) as an opt-in extension of the C ABI.

No reference code runs these (Loader.cpp:122-143,338-346 are commented out; rayhit.rchit:716,729 pass uv = vec2(0)), so
the oracle's oracle_texture.h is the definition and the HIP path must match it bit for bit like everything else.  The
second half checks that scenes which do not set the fields are untouched by the extension's existence.
"""
import numpy as np
import pytest

import textured
from conftest import rmse

pytestmark = pytest.mark.gpu


def render_both(g, oracle_mod, sc, W, H, spp, env=None):
    import os

    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        with g.Context(0) as ctx:
            ctx.upload_scene(sc)
            ctx.frame_begin(W, H)
            ctx.render(spp=spp)
            img = ctx.download().reshape(-1, 4)
            st = ctx.stats()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    ref, ost = oracle_mod.Oracle(sc).render(W, H, spp=spp)
    return img, ref, st, ost


@pytest.mark.parametrize("mode", ["default", "wavefront-only"])
@pytest.mark.parametrize("what", ["textures", "envmap", "both", "both-srgb"])
def test_open_scene_matches_oracle(oracle_mod, what, mode):
    import gpuspectral_amd as g

    sc = textured.decorate(textured.open_scene(16), seed=3, textures=what != "envmap", envmap=what != "textures",
                           decode=textured.srgb_table() if what == "both-srgb" else None)
    img, ref, st, ost = render_both(g, oracle_mod, sc, 96, 64, 4, {"GSP_FINISH_PATHS": "0"} if mode == "wavefront-only" else None)
    assert np.array_equal(img, ref), "RMSE %.3e, %d pixels differ" % (rmse(img, ref), int((img != ref).any(1).sum()))
    assert st["extension_rays"] == ost["extension_rays"] and st["shadow_rays"] == ost["shadow_rays"]


def test_extension_changes_the_image_and_only_where_it_should(oracle_mod):
    """Same scene with and without the fields: the textured spheres / the sky change, and a scene whose has_texture
    fields are set but which passes no textures array renders exactly like the plain one (the reference's case: the
    loader never fills them)."""
    import gpuspectral_amd as g

    plain = textured.open_scene(16)
    img0, ref0, _, _ = render_both(g, oracle_mod, plain, 96, 64, 2)
    assert np.array_equal(img0, ref0)
    flagged = textured.open_scene(16)
    for recs in flagged.bsdfs:
        if "has_texture" in (recs.dtype.names or ()):
            recs["has_texture"] = 7  # set, but no textures array: ignored like the reference's shaders ignore it
    imgf, reff, _, _ = render_both(g, oracle_mod, flagged, 96, 64, 2)
    assert np.array_equal(imgf, img0) and np.array_equal(reff, ref0)
    deco = textured.decorate(textured.open_scene(16), seed=3)
    img1, ref1, _, _ = render_both(g, oracle_mod, deco, 96, 64, 2)
    assert np.array_equal(img1, ref1)
    assert not np.array_equal(img0, img1)
    sky = img0.reshape(64, 96, 4)[:8]  # top rows look at the sky: black without an environment map
    assert float(sky[..., :3].max()) == 0.0 and float(img1.reshape(64, 96, 4)[:8, :, :3].min()) > 0.0


def test_cornell_materials_with_textures(oracle_mod, materials_scene):
    """All eight BSDF types in one closed room, the three texturable ones textured."""
    import copy

    import gpuspectral_amd as g

    sc = textured.decorate(copy.deepcopy(materials_scene), seed=5, envmap=False)
    img, ref, st, ost = render_both(g, oracle_mod, sc, 80, 80, 3)
    assert np.array_equal(img, ref), "RMSE %.3e" % rmse(img, ref)
    assert st["shadow_rays"] == ost["shadow_rays"]


def test_textured_scene_follows_a_refit(oracle_mod, materials_scene):
    """gsp_update_instances on a textured scene (ABI 6): a refit keeps the triangle slots, so the per-slot texture coordinates
    stay where they are; a rebuild gathers them again.  Both frames equal the oracle on the edited scene."""
    import copy

    import gpuspectral_amd as g
    from gpuspectral_amd import abi

    W, H, SPP = 80, 80, 3
    sc = textured.decorate(copy.deepcopy(materials_scene), seed=9)
    inst = sc.instances.copy()
    k = int(np.argmax(inst["vertex_count"]))
    t = inst["transform"][k].copy()
    t[12] += np.float32(0.03)
    t[13] -= np.float32(0.02)
    inst["transform"][k] = t
    frames = []
    for growth in (0.0, 1.0):
        with g.Context(0, options=abi.CtxOptions(refit_growth=growth)) as ctx:
            ctx.upload_scene(sc)
            ctx.frame_begin(W, H)
            ctx.render(spp=1)  # (the pipeline has run on the tree as built)
            ctx.update_instances(inst)
            assert ctx.stats()["scene_refits"] == (1 if growth == 0.0 else 0)
            ctx.frame_begin(W, H)
            ctx.render(spp=SPP)
            frames.append(ctx.download().reshape(-1, 4).copy())
    sc.instances = inst
    ref, _ = oracle_mod.Oracle(sc).render(W, H, spp=SPP)
    assert np.array_equal(frames[0], ref) and np.array_equal(frames[1], ref)


def test_bad_extension_inputs_are_rejected(materials_scene):
    import copy

    import gpuspectral_amd as g
    from gpuspectral_amd import abi

    with g.Context(0) as ctx:
        sc = textured.decorate(copy.deepcopy(materials_scene), seed=1, envmap=False)
        sc.bsdfs[abi.BSDF_NAMES.index("diffuse")]["has_texture"][0] = 99  # beyond the textures array
        with pytest.raises(g.GspError, match="has_texture"):
            ctx.upload_scene(sc)
        sc = textured.decorate(copy.deepcopy(materials_scene), seed=1, envmap=False)
        sc.textures["first_texel"][-1] = len(sc.texels)  # runs off the texel array
        with pytest.raises(g.GspError, match="texture 3"):
            ctx.upload_scene(sc)
        ctx.upload_scene(materials_scene)  # the context is still usable


def test_cpp_host_renders_a_scene_file_with_dormant_features(tmp_path, oracle_mod):
    """End to end through the C++ host: scene.xml + OBJ + PNG / JPEG / PFM files -> loadScene(dormantFeatures) ->
    flattenScene -> PathTracer -> gsp_render, bit-equal to the oracle on the arrays the loader produced; the default
    load of the same file renders the reference's image (no textures, no sky)."""
    pytest.importorskip("PIL.Image")
    from gpuspectral_amd import host

    xml = textured.write_dormant_scene(str(tmp_path))
    W, H, spp = 96, 72, 3
    for dormant in (True, False):
        scene = host.Scene(xml, dormant_features=dormant)
        pt = host.PathTracer(W, H)
        pt.render(scene, spp)
        img = pt.download().reshape(-1, 4)
        pt.close()
        sc = scene.arrays()
        assert (len(sc.textures) == 3 and sc.env_texels is not None) if dormant else (len(sc.textures) == 0 and sc.env_texels is None)
        ref, _ = oracle_mod.Oracle(sc).render(W, H, spp=spp)
        assert np.array_equal(img, ref), "dormant=%s: RMSE %.3e" % (dormant, rmse(img, ref))
        black = float((img[:, :3].max(axis=1) == 0.0).mean())  # pixels whose every path escaped unlit
        assert black < 0.05 if dormant else black > 0.1, black  # (the sky's sun texel exceeds the firefly cutoff: dropped samples)


def test_cli_dormant_features_flag(tmp_path, oracle_mod):
    import os
    import subprocess

    pytest.importorskip("PIL.Image")
    from gpuspectral_amd import host

    xml = textured.write_dormant_scene(str(tmp_path))
    exe = os.path.join(os.path.dirname(host.lib_path()), "gsp_render")
    out = str(tmp_path / "o.pfm")
    r = subprocess.run([exe, "--dormant-features", xml, out, "64", "48", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    got = host.load_hdr_bitmap(out)[::-1, :, :3]  # PFM rows are bottom-up
    ref, _ = oracle_mod.Oracle(host.Scene(xml, dormant_features=True).arrays()).render(64, 48, spp=2)
    assert np.array_equal(got, ref.reshape(48, 64, 4)[..., :3])


def test_multi_gpu_shares_with_textures(oracle_mod):
    """The tiled multi-share path (gsp_multi_*) replicates the extension's arrays like the rest of the scene."""
    import gpuspectral_amd as g
    from gpuspectral_amd import pt

    sc = textured.decorate(textured.open_scene(12), seed=9)
    W, H, spp = 96, 64, 2
    with pt.MultiContext([0, 0, 0]) as m:
        m.upload_scene(sc)
        m.frame_begin(W, H)
        m.render(spp=spp)
        img = m.download().reshape(-1, 4)
    ref, _ = oracle_mod.Oracle(sc).render(W, H, spp=spp)
    assert np.array_equal(img, ref)


def test_random_scenes_with_the_extension_match_oracle(oracle_mod):
    """tests/tools/fuzz_parity.py with `dormant`: random scenes (all eight BSDF types at ordinary and extreme
    parameters) + random uv / textures / environment map / sRGB table, half of them with an open wall.  2 000 more seeds
    were run by hand on the final build (profiles/r02_fuzz.txt): 0 differences."""
    import os
    import sys

    import gpuspectral_amd as g

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "tools"))
    import fuzz_parity

    with g.Context(0) as ctx:
        for seed in range(500, 530):
            ok, ndiff, tris = fuzz_parity.check(ctx, oracle_mod, seed, dormant=True)
            assert ok, "seed %d: %d pixels differ (%d triangles)" % (seed, ndiff, tris)


@pytest.mark.parametrize("env_size", [(1, 1), (2, 1), (5, 3), (64, 32)])
def test_environment_only_scene_and_axis_directions(oracle_mod, env_size):
    """No geometry at all: every primary ray escapes, the image IS the environment lookup -- with cameras that look along
    the lat-long singularities (straight up / down: atan2(0, 0)), along -z (the u = 0.5 seam partner) and along +z (the
    wrap seam), a non-orthonormal world-to-envmap matrix, and degenerate map sizes."""
    import gpuspectral_amd as g
    from gpuspectral_amd import abi

    rng = np.random.RandomState(11)
    w, h = env_size
    looks = {
        "-z": np.eye(4), "+z": np.diag([-1.0, 1.0, -1.0, 1.0]),
        "up": np.array([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1.0]]),
        "down": np.array([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1.0]]),
    }
    for name, m in looks.items():
        sc = abi.SceneArrays()
        sc.to_world = m.T.reshape(16).astype(np.float32)
        sc.fov = np.float32(1.2)
        sc.env_texels = rng.uniform(0.0, 3.0, (h, w, 4)).astype(np.float32)
        shear = np.eye(4)
        shear[0, 1], shear[2, 2] = 0.3, 1.7  # not a rotation: directions are not re-normalised, by definition
        sc.env_to_local = (shear if name == "+z" else np.eye(4)).T.reshape(16).astype(np.float32)
        with g.Context(0) as ctx:
            ctx.upload_scene(sc)
            ctx.frame_begin(33, 33)  # odd size: the centre pixel looks exactly along the axis
            ctx.render(spp=2)
            img = ctx.download().reshape(-1, 4)
            st = ctx.stats()
        ref, ost = oracle_mod.Oracle(sc).render(33, 33, spp=2)
        assert np.array_equal(img, ref), (name, env_size)
        assert st["extension_rays"] == ost["extension_rays"] == 33 * 33 * 2 and st["shadow_rays"] == 0
        assert np.isfinite(img).all() and img[:, :3].max() > 0


def test_texture_coordinate_edge_cases(oracle_mod):
    """One textured quad in front of the camera whose uv take values a file can carry but a sampler must survive: NaN,
    +-inf, 1e30, negative, exactly 0 / 1 / integers (texel-centre rule at the wrap seam), with 1x1, 2x2 and 3x1 textures."""
    import gpuspectral_amd as g
    from gpuspectral_amd import scenes

    rng = np.random.RandomState(2)
    uv_sets = [
        [(0, 0), (1, 0), (0, 1), (0, 1), (1, 0), (1, 1)],
        [(-3.25, 7.5), (2, -1), (0.5, 0.5), (0.5, 0.5), (2, -1), (1e-9, 1 - 1e-7)],
        [(np.nan, 0.2), (0.3, np.inf), (-np.inf, 0.1), (1e30, -1e30), (0.25, 0.75), (3e9, 0.5)],
    ]
    for size in ((1, 1), (2, 2), (3, 1)):
        for uvs in uv_sets:
            b = scenes.SceneBuilder()
            rect = b.add_mesh(*scenes.rect_mesh())
            b.add_object(rect, scenes.rowmajor([2, 0, 0, 0, 0, 2, 0, 0, 0, 0, 1, -3, 0, 0, 0, 1]), b.diffuse((0.5, 0.5, 0.5)), emission=(0, 0, 0))
            b.add_object(rect, scenes.rowmajor([0.5, 0, 0, 0, 0, 0, -1, 2.5, 0, 0.5, 0, -2, 0, 0, 0, 1]), b.diffuse((0, 0, 0)), emission=(20, 20, 20))
            b.camera_lookat((0, 0, 2), (0, 0, -3), fov_deg=50.0)
            sc = b.build()
            sc.uvs = np.zeros((len(sc.positions), 2), np.float32)
            sc.uvs[:6] = np.array(uvs, np.float32)
            sc.add_texture(rng.randint(0, 256, (size[1], size[0], 4)).astype(np.uint8))
            sc.bsdfs[0]["has_texture"][0] = 1
            with g.Context(0) as ctx:
                ctx.upload_scene(sc)
                ctx.frame_begin(48, 48)
                ctx.render(spp=2)
                img = ctx.download().reshape(-1, 4)
            ref, _ = oracle_mod.Oracle(sc).render(48, 48, spp=2)
            assert np.array_equal(img, ref, equal_nan=True), (size, uvs[0])
            assert np.isfinite(img).all() and img[:, :3].max() > 0.01  # the quad is lit and seen


def test_textures_without_uvs_are_rejected(materials_scene):
    import copy

    import gpuspectral_amd as g

    sc = textured.decorate(copy.deepcopy(materials_scene), seed=1, envmap=False)
    d = sc.desc()
    d.uvs = None
    with g.Context(0) as ctx:
        L = g.pt.load()
        import ctypes as C

        assert L.gsp_upload_scene(ctx._h, C.byref(d)) != 0
        assert b"uvs" in L.gsp_last_error(ctx._h)


def test_bench_scene_with_the_extension_at_full_size(oracle_mod):
    """The headline workload (988 k triangles, 1920x1080) with every texturable record textured from 1024^2 / 512^2 images:
    the full frame renders (k_shade<true> over 50 M-path pools), and 256 seeded pixels -- rendered as a pixel subset, which
    reproduces the full frame's values (test_pixel_subset_reproduces_full_frame) -- equal the oracle bit for bit."""
    import zlib

    import gpuspectral_amd as g
    from gpuspectral_amd import abi, scenes

    sc = scenes.interior(1_000_000)
    rng = np.random.RandomState(21)
    sc.uvs = rng.uniform(-1.0, 3.0, (len(sc.positions), 2)).astype(np.float32)
    for size in (1024, 512, 333):
        sc.add_texture(rng.randint(30, 256, (size, size, 4)).astype(np.uint8))
    for name in ("diffuse", "rough_conductor", "rough_plastic"):
        recs = sc.bsdfs[abi.BSDF_NAMES.index(name)]
        recs["has_texture"] = 1 + (np.arange(len(recs)) % 3)
    sc.texel_decode = textured.srgb_table()
    W, H, spp = 1920, 1080, 16
    ids = np.sort(rng.choice(W * H, 256, replace=False)).astype(np.uint32)
    with g.Context(0) as ctx:
        ctx.upload_scene(sc)
        ctx.frame_begin(W, H)
        ctx.render(spp=spp)
        full = ctx.download().reshape(-1, 4)
        st = ctx.stats()
        ctx.frame_begin(W, H, ids)
        ctx.render(spp=spp)
        sub = ctx.download_compact()
    assert np.isfinite(full).all() and st["samples"] == W * H * spp
    assert np.array_equal(sub, full[ids])
    ref, _ = oracle_mod.Oracle(sc).render(W, H, spp=spp, pixel_ids=ids)
    assert np.array_equal(sub, ref), "RMSE %.3e" % rmse(sub, ref)
    assert zlib.crc32(full.tobytes()) != 0
