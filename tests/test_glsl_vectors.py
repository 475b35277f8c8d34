"""The reference's own shader text, executed (tests/golden/glsl_vectors.npz, made by tests/golden/make_glsl_vectors.py from
S/assets/shaders/pt_common.glsl:30-42,86-151 and rayhit.rchit:17-70,89-654 compiled as C++ against oracle/glsl_shim.h), against
  * the CPU oracle's restatement (oracle/oracle_bsdf.h), and
  * the PRODUCT's device headers (gpuspectral_amd/csrc/pt_shading.h, through tests/emu),
bit for bit: 12 800 (record, wo, seed) -> sampleBSDF, 12 800 evalBSDF, 3 000 sampleLight, 2 000 Onb frames, 1 000 RNG rows.

What this is and is not (oracle/README.md, "The reference's own text, executed"): the fixture is not an output of the
reference -- the built-ins' arithmetic (sin / cos / log / exp / normalize / dot) is the shim's, i.e. the oracle's, choice --
so parity stays "unpinned" in the grading sense.  It replaces the human reading of 566 lines of shader by a compiler's:
operator grouping, branch structure, literal values, argument order and the order of random draws are the reference's text.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

FIX = os.path.join(GOLDEN, "glsl_vectors.npz")


@pytest.fixture(scope="module")
def vec():
    with np.load(FIX) as z:
        return {k: np.ascontiguousarray(z[k]) for k in z.files}  # (an NpzFile decompresses on every access)


@pytest.fixture(scope="module")
def tables(vec):
    """The fixture's own record tables as a scene (so a change of make_glsl_vectors.tables() cannot go unnoticed)."""
    sys.path.insert(0, GOLDEN)
    from make_glsl_vectors import tables as mk

    sc = mk()
    for i in range(8):
        assert np.array_equal(sc.bsdfs[i].view(np.uint8), vec["bsdf%d" % i].view(np.uint8))
    assert np.array_equal(sc.lights.view(np.uint8), vec["light_table"].view(np.uint8))
    return sc


def _first_bad(a, b):
    bad = np.nonzero((a != b).any(axis=1))[0]
    return None if not len(bad) else int(bad[0])


def _check_bsdf(side, vec, what):
    n = len(vec["handles"])
    got_s = np.zeros((n, 9), np.uint32)
    got_e = np.zeros((n, 5), np.uint32)
    for i in range(n):
        h, s = int(vec["handles"][i]), int(vec["seeds"][i])
        got_s[i] = side.bsdf_sample(h, vec["wo"][i], s).view(np.uint32)
        got_e[i] = side.bsdf_eval(h, vec["wo"][i], vec["wi"][i]).view(np.uint32)
    i = _first_bad(got_s, vec["samples"])
    assert i is None, "%s sampleBSDF differs from the executed shader text at vector %d: handle %#x wo %s seed %d\n got %s\nwant %s" % (
        what, i, vec["handles"][i], vec["wo"][i], vec["seeds"][i], got_s[i].view(np.float32), vec["samples"][i].view(np.float32))
    i = _first_bad(got_e, vec["evals"])
    assert i is None, "%s evalBSDF differs at vector %d: handle %#x wo %s wi %s\n got %s\nwant %s" % (
        what, i, vec["handles"][i], vec["wo"][i], vec["wi"][i], got_e[i].view(np.float32), vec["evals"][i].view(np.float32))
    assert set(vec["handles"] >> 16) == set(range(8)) and n >= 10000


def _check_lights(side, vec, what):
    n = len(vec["light_pos"])
    got = np.zeros((n, 8), np.uint32)
    for i in range(n):
        got[i] = side.sample_light(vec["light_pos"][i], int(vec["light_seeds"][i])).view(np.uint32)
    i = _first_bad(got, vec["lights"])
    assert i is None, "%s sampleLight differs at %d: pos %s seed %d\n got %s\nwant %s" % (
        what, i, vec["light_pos"][i], vec["light_seeds"][i], got[i].view(np.float32), vec["lights"][i].view(np.float32))


def test_fixture_covers_what_it_claims(vec):
    s = vec["samples"].view(np.float32)
    t = vec["handles"] >> 16
    assert len(t) >= 10000 and all((t == k).sum() >= 1000 for k in range(8))
    # both lobes of every two-lobe model, both outcomes of the dielectric, total internal reflection, non-finite outputs
    for k in (3, 5):  # smooth plastic / floor: delta reflection and diffuse
        d = s[t == k, 7]
        assert 0.02 < d.mean() < 0.98
    die = t == 1
    assert ((s[die, 2] * vec["wo"][die, 2]) < 0).sum() > 300 and ((s[die, 2] * vec["wo"][die, 2]) > 0).sum() > 100
    assert (~np.isfinite(s[:, :7])).any()  # NaN / inf rows exist and must agree too
    assert (vec["wi"] == s[:, :3]).all(axis=1).mean() > 0.3  # evalBSDF at the sampled direction (the MIS use)
    assert len(vec["lights"]) >= 3000 and len(vec["onb"]) >= 2000 and len(vec["rng"]) >= 1000


def test_oracle_equals_the_executed_shader_text(oracle_mod, vec, tables):
    o = oracle_mod.Oracle(tables)
    _check_bsdf(o, vec, "oracle")
    _check_lights(o, vec, "oracle")
    L = oracle_mod.lib()
    L.oracle_onb.argtypes = [C.c_void_p] * 3
    L.oracle_power_heuristic.restype = C.c_float
    L.oracle_power_heuristic.argtypes = [C.c_float, C.c_float]
    L.oracle_cosine_pdf.restype = C.c_float
    L.oracle_cosine_pdf.argtypes = [C.c_float]
    L.oracle_is_transmission.argtypes = [C.c_uint32]
    out = np.zeros(15, np.float32)
    for i in range(len(vec["onb"])):
        n, v = vec["onb_n"][i].copy(), vec["onb_v"][i].copy()
        L.oracle_onb(n.ctypes.data, v.ctypes.data, out.ctypes.data)
        assert np.array_equal(out.view(np.uint32), vec["onb"][i]), (i, vec["onb_n"][i])
    for i in range(len(vec["rng"])):
        a, b = int(vec["rng_a"][i]), int(vec["rng_b"][i])
        draws, final = oracle_mod.rand_pcg(a, 2)
        u = np.float32(oracle_mod.rand_uniform(int(final)))
        _, after = oracle_mod.rand_pcg(int(final), 1)
        got = [oracle_mod.tea(a, b), oracle_mod.pcg_hash(a), int(draws[0]), int(draws[1]), int(u.view(np.uint32)), int(after)]
        assert got == [int(x) for x in vec["rng"][i]], (i, a, b)
    for i in range(len(vec["helpers"])):
        f, g, h = float(vec["helper_f"][i]), float(vec["helper_g"][i]), int(vec["helper_handles"][i])
        got = [np.float32(L.oracle_power_heuristic(f, g)).view(np.uint32), np.float32(L.oracle_cosine_pdf(f)).view(np.uint32),
               L.oracle_is_transmission(h)]
        assert [int(x) for x in got] == [int(x) for x in vec["helpers"][i]], (i, f, g)


def test_product_headers_equal_the_executed_shader_text(emu, vec, tables):
    e = emu.scene(tables)
    _check_bsdf(e, vec, "pt_shading.h")
    _check_lights(e, vec, "pt_shading.h")
    L = emu.L
    L.emu_onb.argtypes = [C.c_void_p] * 3
    L.emu_rng.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
    L.emu_power_heuristic.restype = C.c_float
    L.emu_power_heuristic.argtypes = [C.c_float, C.c_float]
    L.emu_cosine_pdf.restype = C.c_float
    L.emu_cosine_pdf.argtypes = [C.c_float]
    L.emu_is_transmission.argtypes = [C.c_uint32]
    out = np.zeros(15, np.float32)
    for i in range(len(vec["onb"])):
        n, v = vec["onb_n"][i].copy(), vec["onb_v"][i].copy()
        L.emu_onb(n.ctypes.data, v.ctypes.data, out.ctypes.data)
        assert np.array_equal(out.view(np.uint32), vec["onb"][i]), (i, vec["onb_n"][i])
    r = np.zeros(6, np.uint32)
    for i in range(len(vec["rng"])):
        L.emu_rng(int(vec["rng_a"][i]), int(vec["rng_b"][i]), r.ctypes.data)
        assert np.array_equal(r, vec["rng"][i]), i
    for i in range(len(vec["helpers"])):
        f, g, h = float(vec["helper_f"][i]), float(vec["helper_g"][i]), int(vec["helper_handles"][i])
        got = [np.float32(L.emu_power_heuristic(f, g)).view(np.uint32), np.float32(L.emu_cosine_pdf(f)).view(np.uint32),
               L.emu_is_transmission(h)]
        assert [int(x) for x in got] == [int(x) for x in vec["helpers"][i]], (i, f, g)


def _image_cases(vec):
    sys.path.insert(0, GOLDEN)
    import make_glsl_vectors as M

    for name, (w, h, spp) in M.IMAGES.items():
        yield name, M.image_scene(name), w, h, spp, vec["img_" + name].reshape(-1, 4), vec["img_" + name + "_rays"]


def test_oracle_image_equals_the_executed_shaders(oracle_mod, vec):
    """Second step (r06): not only the pure functions but the shaders' main() functions -- raygen.rgen:29-108 (camera ray, bounce
    loop, firefly cutoff, Russian roulette, running mean), rayhit.rchit:666-797 (the closest-hit shader: normals, twofaced flip, Onb,
    sampleBSDF / sampleLight / evalBSDF, the NEE branch with its shadow ray, MIS weights, the termination tests, the payload
    update), miss.rmiss, shadowmiss.rmiss -- were compiled from the reference's text and run for whole frames; traceRayEXT's
    traversal is the oracle's (the driver's is vendor-opaque, oracle/README.md).  The oracle's restatement of those lines must produce
    the same frame, bit for bit, and trace the same number of extension and shadow rays.  Cornell box (the reference's scene.xml)
    96 x 96 x 8 spp; Cornell + all eight BSDF types (glass: delta paths to depth 50) 96 x 80 x 4 spp; 24 random scenes of the parity
    fuzzer (extreme BSDF parameters, mirrored / sheared transforms) 40 x 28 x 3 spp: 1.07 M rays through the executed shaders."""
    for name, sc, w, h, spp, img, rays in _image_cases(vec):
        got, st = oracle_mod.Oracle(sc).render(w, h, spp=spp)
        assert np.array_equal(got.view(np.uint32), img.view(np.uint32)), (name, int((got != img).any(axis=1).sum()))
        assert (st["extension_rays"], st["shadow_rays"]) == (int(rays[0]), int(rays[1])), name
        if not name.startswith("fuzz"):  # (the two lit rooms; a random scene may be dark or all-delta)
            assert np.isfinite(img).all() and img[:, :3].mean() > 0.05 and rays[1] > 0.3 * rays[0]


def test_product_stage_headers_image_equals_the_executed_shaders(emu, vec):
    """... and so must the product's own per-path stage code (pt_stages.h / pt_shading.h / pt_trace.h compiled for the host)."""
    for name, sc, w, h, spp, img, _ in _image_cases(vec):
        got = emu.scene(sc).render(w, h, spp=spp)
        assert np.array_equal(got.view(np.uint32), img.view(np.uint32)), (name, int((got != img).any(axis=1).sum()))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is mounted in the build container only")
def test_fixture_is_what_the_reference_text_produces_today():
    """Where /root/reference exists: rebuild oracle/_ref/libglsl_ref.so from the shader files where they lie and regenerate
    every vector -- the committed fixture must be reproduced byte for byte."""
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_glsl_vectors.py"), "--check"], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "identical" in r.stdout
