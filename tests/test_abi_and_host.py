"""C-ABI surface, struct layouts, loud failure without a GPU, and the C++ host layer
(Mitsuba loader / Scene flattening / PFM writer) against the numpy restatement."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import CORNELL_XML, ROOT, has_gpu

HEADER = os.path.join(ROOT, "include", "gpuspectral_pt.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gsp_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import gpuspectral_amd as g
    from gpuspectral_amd import pt

    L = pt.load()
    syms = declared_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(L, s), "libgpuspectral_pt.so does not export %s" % s
    assert sorted(pt.EXPORTS) == syms
    assert L.gsp_abi_version() == g.abi.GSP_ABI_VERSION


def test_loaded_library_was_built_from_this_tree():
    """gsp_build_info(): the digest stamped into the .so by csrc/Makefile == the digest of the sources in the tree, so a
    stale prebuilt library cannot stand in for the code under review; the flags that pin the float arithmetic are on."""
    from gpuspectral_amd import pt

    if os.environ.get("GSP_LIB_PATH"):
        pytest.skip("GSP_LIB_PATH points at a build variant")
    info = pt.build_info()
    assert info["arch"] == "gfx950"
    assert info["digest"] == pt.source_digest(), "libgpuspectral_pt.so is stale: run make -C gpuspectral_amd/csrc"
    assert "-ffp-contract=off" in info["flags"].split()


def test_valu_mix_record_belongs_to_the_loaded_library():
    """csrc/Makefile writes lib/valu_mix.json (static VALU class mix of the three render kernels, scripts/valu_mix.py) next to
    the library with the library's source digest; bench.py turns it into the class-weighted issue ceiling of `roofline` and
    ignores a record of another build."""
    import json

    from gpuspectral_amd import pt

    if os.environ.get("GSP_LIB_PATH"):
        pytest.skip("GSP_LIB_PATH points at a build variant")
    rec = json.load(open(os.path.join(os.path.dirname(pt.lib_path()), "valu_mix.json")))
    assert rec["library_digest"] == pt.build_info()["digest"]
    for k in ("k_trace_extend", "k_trace_connect", "k_shade"):
        m = rec["kernels"][k]
        assert m["full"] + m["half"] + m["quarter"] == m["valu_instructions_static"] > 300
        assert 2.0 < m["mean_issue_cycles"] < 4.0


def test_struct_layouts_match_header(tmp_path):
    """sizeof/offsetof from the C header (gcc) == the ctypes/numpy mirrors."""
    from gpuspectral_amd import abi

    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "gpuspectral_pt.h"\n'
        "int main(){printf(\"%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n\","
        "sizeof(gsp_diffuse_bsdf),sizeof(gsp_smooth_dielectric_bsdf),sizeof(gsp_smooth_conductor_bsdf),"
        "sizeof(gsp_smooth_plastic_bsdf),sizeof(gsp_rough_conductor_bsdf),sizeof(gsp_smooth_floor_bsdf),"
        "sizeof(gsp_rough_floor_bsdf),sizeof(gsp_rough_plastic_bsdf),sizeof(gsp_triangle_light),sizeof(gsp_instance),"
        "sizeof(gsp_scene_desc),sizeof(gsp_render_params),sizeof(gsp_stats),offsetof(gsp_scene_desc,camera),"
        "offsetof(gsp_scene_desc,num_bsdfs));"
        "printf(\"%zu %zu %zu %zu %zu %zu\\n\",sizeof(gsp_ctx_options),offsetof(gsp_ctx_options,memory_share),"
        "offsetof(gsp_ctx_options,gather_route),offsetof(gsp_render_params,disable_nee),offsetof(gsp_stats,scene_updates),sizeof(gsp_camera));"
        "return 0;}\n"
    )
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    vals = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert tuple(vals[:8]) == abi.BSDF_SIZES == (16, 8, 8, 24, 44, 16, 20, 32)  # S/renderer/Scene.h:29-81
    assert vals[8] == abi.LIGHT_DT.itemsize == 64
    assert vals[9] == abi.INSTANCE_DT.itemsize
    assert vals[10] == C.sizeof(abi.SceneDesc)
    assert vals[11] == C.sizeof(abi.RenderParams)
    assert vals[12] == C.sizeof(abi.Stats)
    assert vals[13] == abi.SceneDesc.camera.offset and vals[14] == abi.SceneDesc.num_bsdfs.offset
    # ABI 5
    assert vals[15] == C.sizeof(abi.CtxOptions) and vals[16] == abi.CtxOptions.memory_share.offset
    assert vals[17] == abi.CtxOptions.gather_route.offset
    assert vals[18] == abi.RenderParams.disable_nee.offset and vals[19] == abi.Stats.scene_updates.offset
    assert vals[20] == C.sizeof(abi.Camera)


def test_default_params_are_the_reference_literals():
    from gpuspectral_amd import abi, pt

    p = abi.RenderParams()
    pt.load().gsp_default_render_params(C.byref(p))
    assert (p.spp, p.first_timestamp, p.max_depth, p.rr_start_depth, p.clamp) == (1, 0, 50, 10, 20.0)
    assert p.disable_nee == 0  # `#define NEE true`, rayhit.rchit:656
    assert p.struct_size == C.sizeof(abi.RenderParams) and abi.RenderParams.struct_size.offset == 0  # (ABI 8: first field)
    q = abi.default_render_params()
    assert q.struct_size == C.sizeof(abi.RenderParams)
    assert (q.max_depth, q.rr_start_depth, q.clamp, q.disable_nee) == (50, 10, 20.0, 0)


def test_default_ctx_options_and_env_mapping():
    """gsp_default_ctx_options fills the documented defaults; the TEST binding's env mapping lands in the right fields."""
    from gpuspectral_amd import abi, pt

    o = abi.CtxOptions()
    pt.load().gsp_default_ctx_options(C.byref(o))
    assert o.struct_size == C.sizeof(abi.CtxOptions)
    assert (o.lanes, o.pool_paths, o.ring_bytes, o.primary_memo, o.finish_paths, o.reinsert_rounds, o.gather_route) == \
        (1, 96 << 20, 16 << 30, 1, 262144, 7, abi.GATHER_AUTO)
    assert o.memory_share == 0.4
    e = abi.options_from_env({"GSP_POOL_PATHS": "4000000", "GSP_PRIMARY_MEMO": "0", "GSP_FINISH_PATHS": "0", "GSP_LANES": "2",
                              "GSP_BVH_REINSERT": "0", "GSP_MULTI_GATHER": "rccl", "GSP_MEMORY_SHARE": "0.1"})
    assert (e.pool_paths, e.primary_memo, e.finish_paths, e.lanes, e.reinsert_rounds, e.gather_route, e.memory_share) == \
        (4000000, 2, 0xFFFFFFFF, 2, 1, abi.GATHER_RCCL, 0.1)
    z = abi.options_from_env({})
    assert (z.pool_paths, z.primary_memo, z.lanes) == (0, 0, 0) and z.struct_size == C.sizeof(abi.CtxOptions)  # 0 = default


def test_library_reads_no_environment_and_links_no_rccl():
    """VERDICT r03 weak 7 / ADVICE: behaviour switches are gsp_ctx_options, not getenv() inside a C-ABI library; librccl is
    resolved with dlopen at the first multi-GPU create, so the library loads where RCCL is absent."""
    from gpuspectral_amd import pt

    csrc = os.path.join(ROOT, "gpuspectral_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h", ".inc")):
            txt = open(os.path.join(csrc, f)).read()
            assert "getenv" not in txt, f
            assert "GSP_WIDE" not in txt, f  # the rejected 8-wide variant is history, not a macro in the hot path
    needed = subprocess.check_output(["readelf", "-d", pt.lib_path()], text=True)
    assert "librccl" not in needed
    host_dir = os.path.join(ROOT, "gpuspectral_amd", "host")
    for f in os.listdir(host_dir):
        if f.endswith((".cpp", ".h")):
            assert "getenv" not in open(os.path.join(host_dir, f)).read(), f


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly_no_fallback():
    import gpuspectral_amd as g
    from gpuspectral_amd import host

    assert g.device_count() == 0
    with pytest.raises(g.GspError, match="no CPU fallback"):
        g.Context(0)
    with pytest.raises(g.GspError):
        host.PathTracer(16, 16)


def test_product_never_imports_the_oracle():
    """The package, its C++ sources and the timed part of bench.py must not reference oracle/."""
    pkg = os.path.join(ROOT, "gpuspectral_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f
                assert not re.search(r'#include\s+"[^"]*oracle', txt), f  # comments may cite it, code may not use it
    # scripts/ (measurement helpers) never touch it either; checkers that need it live under tests/tools/
    for f in os.listdir(os.path.join(ROOT, "scripts")):
        if f.endswith((".py", ".sh")):
            txt = open(os.path.join(ROOT, "scripts", f), errors="replace").read()
            assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f
    # bench.py: only inside cpu_baseline(); __graft_entry__.py: only inside smoke()
    import ast

    for fname, allowed in (("bench.py", "cpu_baseline"), ("__graft_entry__.py", "smoke")):
        tree = ast.parse(open(os.path.join(ROOT, fname)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.FunctionDef):
                uses = [n for n in ast.walk(node) if isinstance(n, (ast.Import, ast.ImportFrom)) and
                        any("oracle" in (a.name or "") for a in n.names) or
                        (isinstance(n, ast.ImportFrom) and n.module and "oracle" in n.module)]
                assert not uses or node.name == allowed, "%s: %s imports the oracle" % (fname, node.name)
        top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) and
               (any("oracle" in a.name for a in n.names) or (isinstance(n, ast.ImportFrom) and n.module and "oracle" in n.module))]
        assert not top, fname
    code = "import sys; import gpuspectral_amd, gpuspectral_amd.host, gpuspectral_amd.scenes, gpuspectral_amd.multigpu; " \
           "assert not [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')]"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)


def test_scene_tracker_classifies_edits():
    """PathTracer::prepareScene compares the scene BY VALUE with what it uploaded (host/PathTracer.h SceneTracker): the
    reference re-reads camera, transforms, materials, tables and lights every frame (PathTracer.cpp:58-93).  Pure host logic."""
    from gpuspectral_amd import host

    L = host.load()
    NONE, CAM, TAB, INST, ALL = 0, 1, 2, 4, 8
    a = host.Scene(CORNELL_XML)
    assert L.gsph_tracker_probe(a._h, a._h) == NONE
    b = host.Scene(CORNELL_XML)  # the same file loaded again: other Mesh objects => another object list
    assert L.gsph_tracker_probe(a._h, b._h) == ALL
    # edits of ONE scene between remember() and diff(): probe(scene, None) remembers, then the edits, then probe_again
    assert L.gsph_tracker_remember(a._h) == 0
    assert L.gsph_tracker_diff(a._h) == NONE
    arr = a.arrays()
    cam = arr.to_world.copy()
    cam[12] += 0.25
    a.set_camera(cam, float(arr.fov))
    assert L.gsph_tracker_diff(a._h) == CAM
    a.set_diffuse_reflectance(0, [0.1, 0.2, 0.3])
    assert L.gsph_tracker_diff(a._h) == CAM | TAB
    m = arr.instances["transform"][3].copy()
    m[13] += 0.1
    a.set_transform(3, m)
    assert L.gsph_tracker_diff(a._h) == CAM | TAB | INST
    assert L.gsph_tracker_remember(a._h) == 0 and L.gsph_tracker_diff(a._h) == NONE
    a.set_object_material(2, emission=[0.0, 0.0, 0.0], twofaced=False)
    assert L.gsph_tracker_diff(a._h) in (NONE, INST)  # (INST unless object 2 already had these values)
    a.make_object_rough_conductor(5, [0.2, 0.9, 1.1], [3.9, 2.4, 2.2], 0.1)
    assert L.gsph_tracker_diff(a._h) & (TAB | INST) == TAB | INST


# ---- C++ host layer ---------------------------------------------------------------
def same_scene(a, b):
    for name in ("instances", "positions", "normals", "lights", "to_world"):
        x, y = np.asarray(getattr(a, name)), np.asarray(getattr(b, name))
        assert x.shape == y.shape and x.tobytes() == y.tobytes(), name
    assert np.float32(a.fov) == np.float32(b.fov)
    for i, (x, y) in enumerate(zip(a.bsdfs, b.bsdfs)):
        assert len(x) == len(y) and x.tobytes() == y.tobytes(), "bsdf table %d" % i


def test_cpp_loader_matches_numpy_restatement_on_cornell():
    from gpuspectral_amd import host
    from oracle import mitsuba_loader as ml

    s = host.Scene(CORNELL_XML)
    assert s.warnings == [] and s.num_materials == 8
    same_scene(s.arrays(), ml.load_scene(CORNELL_XML))


MATERIALS_XML = """<?xml version="1.0" encoding="utf-8"?>
<!-- build-authored test scene modelled on the reference's test3/scene.xml:56-105,165-178 -->
<scene version="0.5.0">
  <sensor type="perspective">
    <float name="fov" value="19.5"/>
    <transform name="toWorld"><matrix value="-1 0 0 0 0 1 0 1 0 0 -1 6.8 0 0 0 1"/></transform>
  </sensor>
  <bsdf type="twosided" id="White"><bsdf type="diffuse"><rgb name="reflectance" value="0.725, 0.71, 0.68"/></bsdf></bsdf>
  <bsdf type="dielectric" id="Glass"><float name="intIOR" value="1.3"/><float name="extIOR" value="1"/></bsdf>
  <bsdf type="twosided" id="Metal"><bsdf type="roughconductor">
      <float name="alpha" value="0.1"/><string name="distribution" value="ggx"/>
      <rgb name="specularReflectance" value="1, 1, 1"/>
      <rgb name="eta" value="1.65746, 0.880369, 0.521229"/><rgb name="k" value="9.22387, 6.26952, 4.837"/>
  </bsdf></bsdf>
  <bsdf type="twosided" id="Plastic"><bsdf type="roughplastic">
      <float name="alpha" value="0.05"/><float name="intIOR" value="1.3"/><float name="extIOR" value="1"/>
      <rgb name="diffuseReflectance" value="1, 0.578676, 0.134734"/></bsdf></bsdf>
  <bsdf type="twosided" id="Smooth"><bsdf type="plastic"><rgb name="diffuseReflectance" value="0.2 0.3 0.9"/></bsdf></bsdf>
  <bsdf type="conductor" id="Mirror"><string name="material" value="none"/></bsdf>
  <bsdf id="Light"><bsdf type="diffuse"><rgb name="reflectance" value="0"/></bsdf></bsdf>
  <shape type="rectangle"><transform name="toWorld"><matrix value="2 0 0 0 0 0 2 0 0 -2 0 0 0 0 0 1"/></transform><ref id="White"/></shape>
  <shape type="obj"><string name="filename" value="sphere.obj"/>
    <transform name="toWorld"><matrix value="0.4 0 0 0.6 0 0.4 0 0.4 0 0 0.4 -0.1 0 0 0 1"/></transform><ref id="Metal"/></shape>
  <shape type="obj"><string name="filename" value="sphere.obj"/>
    <transform name="toWorld"><matrix value="0.4 0 0 -0.6 0 0.4 0 0.4 0 0 0.4 -0.1 0 0 0 1"/></transform><ref id="Glass"/></shape>
  <shape type="obj"><string name="filename" value="quad.obj"/><point name="center" x="0" y="0.2" z="0.9"/><ref id="Plastic"/></shape>
  <shape type="cube"><transform name="toWorld"><matrix value="0.2 0 0 0 0 0.2 0 0.2 0 0 0.2 -0.8 0 0 0 1"/></transform><ref id="Smooth"/></shape>
  <shape type="cube"><transform name="toWorld"><matrix value="0.3 0 0 0 0 0.3 0 1.0 0 0 0.02 -0.95 0 0 0 1"/></transform><ref id="Mirror"/></shape>
  <shape type="sphere"><float name="radius" value="1"/><ref id="White"/></shape>
  <shape type="rectangle">
    <transform name="toWorld"><matrix value="0.235 0 0 -0.005 0 0 -0.0893 1.98 0 0.19 0 -0.03 0 0 0 1"/></transform>
    <ref id="Light"/><emitter type="area"><rgb name="radiance" value="17, 12, 4"/></emitter></shape>
  <emitter type="envmap"><string name="filename" value="none.hdr"/></emitter>
</scene>
"""


def write_obj(path, pos, nrm, quads=False):
    with open(path, "w") as f:
        for p in pos:
            f.write("v %.9g %.9g %.9g\n" % tuple(p))
        for n in nrm:
            f.write("vn %.9g %.9g %.9g\n" % tuple(n))
        f.write("vt 0 0\n")
        step = 4 if quads else 3
        for i in range(0, len(pos), step):
            f.write("f " + " ".join("%d/1/%d" % (i + k + 1, i + k + 1) for k in range(step)) + "\n")


def test_cpp_loader_matches_numpy_restatement_on_material_scene(tmp_path, oracle_mod):
    """All loader material branches, <ref>, center override, quads, relative OBJ indices, skipped shapes."""
    from gpuspectral_amd import host, scenes
    from oracle import mitsuba_loader as ml

    xml = tmp_path / "scene.xml"
    xml.write_text(MATERIALS_XML)
    pos, nrm = scenes.sphere_mesh(16, 8)
    write_obj(tmp_path / "sphere.obj", pos, nrm)
    q = np.array([(-0.2, 0, -0.2), (0.2, 0, -0.2), (0.2, 0, 0.2), (-0.2, 0, 0.2)], np.float32)
    with open(tmp_path / "quad.obj", "w") as f:  # one quad with negative (relative) indices
        for p in q:
            f.write("v %g %g %g\n" % tuple(p))
        f.write("vn 0 1 0\nvt 0 0\nf -4/-1/-1 -3/-1/-1 -2/-1/-1 -1/-1/-1\n")
    s = host.Scene(str(xml))
    ref = ml.load_scene(str(xml))
    a = s.arrays()
    same_scene(a, ref)
    assert len(a.instances) == 7 and len(a.lights) == 2
    assert any("sphere" in w for w in s.warnings) and any("envmap" in w for w in s.warnings)
    assert [len(b) for b in a.bsdfs] == [2, 1, 1, 1, 1, 0, 0, 1]
    assert np.isclose(a.bsdfs[4]["alpha"][0], np.float32(np.sqrt(2.0)) * np.float32(0.1))  # Loader.cpp:225
    assert a.instances["twofaced"].tolist() == [1, 1, 0, 1, 1, 0, 0]
    assert np.allclose(a.instances["transform"][3][12:15], (0, 0.2, 0.9))  # center override, Loader.cpp:287-293
    # and the scene renders identically through the oracle whichever loader produced it
    i1, _ = oracle_mod.Oracle(a).render(24, 24, spp=2)
    i2, _ = oracle_mod.Oracle(ref).render(24, 24, spp=2)
    assert np.array_equal(i1, i2)


def test_cpp_loader_dormant_features_match_numpy_restatement(tmp_path, oracle_mod):
    """LoadOptions::dormantFeatures.  Off (the default, = the reference): textured slots fall back to the
    colour default with a warning, the envmap emitter is ignored, nothing of the extension is set.  On: the C++ loader
    and the numpy restatement agree array by array (PNG texels exactly, JPEG texels to the decoders' few levels, the
    world-to-envmap matrix to rounding), and the oracle renders the same image from either."""
    pytest.importorskip("PIL.Image")
    import textured
    from gpuspectral_amd import host
    from oracle import mitsuba_loader as ml

    xml = textured.write_dormant_scene(str(tmp_path))
    off = host.Scene(xml)
    a0 = off.arrays()
    same_scene(a0, ml.load_scene(xml))
    assert len(a0.textures) == 0 and a0.uvs is None and a0.env_texels is None
    assert sum("textured" in w for w in off.warnings) == 5 and any("envmap" in w for w in off.warnings)
    assert a0.bsdfs[0]["has_texture"].tolist() == [0, 0, 0] and np.allclose(a0.bsdfs[0]["reflectance"][1], 0.5)

    on = host.Scene(xml, dormant_features=True)
    a, b = on.arrays(), ml.load_scene(xml, dormant_features=True)
    same_scene(a, b)  # instances, geometry, BSDF tables (has_texture values included), lights, camera
    assert a.bsdfs[0]["has_texture"].tolist() == [1, 2, 0]       # checkerboard, paint.png, the emitter's black diffuse
    assert a.bsdfs[7]["has_texture"].tolist() == [3] and a.bsdfs[4]["has_texture"].tolist() == [2]  # wood.jpg; paint.png shared
    assert [w for w in on.warnings if "textured" in w] == ["plastic: textured diffuse_reflectance unsupported (no hasTexture field), colour default used"]
    assert a.textures.tobytes() == b.textures.tobytes() and a.textures["width"].tolist() == [400, 80, 80]
    assert a.uvs.tobytes() == b.uvs.tobytes() and len(a.uvs) == len(a.positions)
    ta, tb = a.texels.view(np.uint8).reshape(-1, 4), b.texels.view(np.uint8).reshape(-1, 4)
    n_exact = 400 * 600 + 80 * 48  # checkerboard + PNG
    assert np.array_equal(ta[:n_exact], tb[:n_exact])
    assert np.abs(ta[n_exact:].astype(int) - tb[n_exact:].astype(int)).max() <= 4  # JPEG: own decoder vs libjpeg
    assert np.array_equal(a.texel_decode, b.texel_decode) and a.texel_decode[0] == 0 and a.texel_decode[255] == 1
    assert np.array_equal(a.env_texels, b.env_texels) and a.env_texels.shape == (16, 32, 4)
    assert np.allclose(a.env_to_local, b.env_to_local, atol=1e-6)
    rot = np.array(a.env_to_local).reshape(4, 4).T  # world -> envmap = inverse of the file's (row-major) toWorld
    assert np.allclose(rot @ np.array([[0.8, 0, 0.6, 0], [0, 1, 0, 0], [-0.6, 0, 0.8, 0], [0, 0, 0, 1]]), np.eye(4), atol=1e-6)
    # the same data through the oracle: identical images once the two loaders' arrays are identical
    b.texels, b.env_to_local = a.texels.copy(), a.env_to_local.copy()
    i1, _ = oracle_mod.Oracle(a).render(40, 30, spp=2)
    i2, _ = oracle_mod.Oracle(b).render(40, 30, spp=2)
    assert np.array_equal(i1, i2)
    i0, _ = oracle_mod.Oracle(a0).render(40, 30, spp=2)
    assert not np.array_equal(i0, i1)
    # a bitmap the scene names but the directory does not hold: warning + colour default (both loaders)
    xml2 = str(tmp_path / "scene_missing.xml")
    with open(xml2, "w") as f:
        f.write(textured.DORMANT_XML.replace("tex/wood.jpg", "tex/absent.jpg"))
    gone = host.Scene(xml2, dormant_features=True)
    ag, bg = gone.arrays(), ml.load_scene(xml2, dormant_features=True)
    same_scene(ag, bg)
    assert ag.bsdfs[7]["has_texture"].tolist() == [0] and len(ag.textures) == 2
    assert any("absent.jpg' not found" in w for w in gone.warnings) and any("absent.jpg' not found" in w for w in bg.warnings)


def test_cpp_loader_errors_are_reported(tmp_path):
    from gpuspectral_amd import host
    from gpuspectral_amd.pt import GspError

    with pytest.raises(GspError, match="cannot open"):
        host.Scene(str(tmp_path / "missing.xml"))
    bad = tmp_path / "bad.xml"
    bad.write_text("<scene><shape type='obj'></scene>")
    with pytest.raises(GspError, match="XML"):
        host.Scene(str(bad))


def test_write_pfm_roundtrip(tmp_path):
    from gpuspectral_amd import host

    img = np.random.RandomState(0).rand(5, 7, 4).astype(np.float32)
    path = str(tmp_path / "o.pfm")
    host.write_pfm(path, img)
    raw = open(path, "rb").read()
    hdr, rest = raw.split(b"\n-1.0\n", 1)
    assert hdr == b"PF\n7 5"
    data = np.frombuffer(rest, "<f4").reshape(5, 7, 3)[::-1]
    assert np.array_equal(data, img[:, :, :3])


REF_SCENES = "/root/reference/src/GPUSpectral/assets/scenes"


def _check_reference_scene(xml, name):
    from gpuspectral_amd import host
    from oracle import mitsuba_loader as ml

    s = host.Scene(xml)
    a, b = s.arrays(), ml.load_scene(xml)
    same_scene(a, b)
    assert a.num_triangles > 10000
    if name != "living-room":  # its only emitter sits on a sphere shape, which the reference loader cannot load
        assert len(a.lights) > 0
    assert sorted(s.warnings) == sorted(b.warnings) or len(s.warnings) >= len(b.warnings)
    print(name, a.num_triangles, "triangles,", len(a.instances), "objects,", len(a.lights), "lights;", len(s.warnings), "warnings")
    return a


def test_cpp_loader_on_staircase2(staircase2_xml):
    """SURVEY 8(f).1 on a committed fixture: the reference's own 'Modern Hall' scene (textures, <ref>, nested
    twosided, roughplastic / dielectric / conductor materials, 16 emitters) loads through the C++ loader, flattens to
    the same arrays as the numpy restatement and to exactly the arrays cached for the GPU tests."""
    from gpuspectral_amd import abi

    a = _check_reference_scene(staircase2_xml, "staircase2")
    assert a.num_triangles == 30927 and len(a.lights) == 16
    cached = abi.SceneArrays.load(os.path.join(os.path.dirname(os.path.dirname(CORNELL_XML)), "ref_scenes", "staircase2.npz"))
    same_scene(a, cached)


@pytest.mark.parametrize("name", ["coffee", "living-room"])
def test_cpp_loader_on_reference_scenes(name):
    """The larger shipped scenes (`conductor material=none`, disk/sphere/hair shapes, missing OBJ blobs), where the
    reference tree is mounted (build container only; staircase2 above runs everywhere)."""
    xml = os.path.join(REF_SCENES, name, "scene.xml")
    if not os.path.exists(xml):
        pytest.skip("reference tree not available")
    _check_reference_scene(xml, name)


def test_builtin_disk_and_sphere_meshes():
    """LoadOptions::builtinShapes: Mitsuba's unit disk and unit sphere, tessellated with float32 + * / sqrt
    only.  Geometry: every rim / surface point on the unit circle / sphere to an ulp, the disk's normal +z and its area
    that of the inscribed 64-gon, the sphere closed (every edge in exactly two triangles, opposite directions) and wound
    outward."""
    from oracle import mitsuba_loader as ml

    pos, nrm, uv = ml.builtin_disk()
    assert pos.shape == (192, 3) and (pos[:, 2] == 0).all() and (nrm == [0, 0, 1]).all()
    rim = pos.reshape(-1, 3, 3)[:, 1:, :]
    assert np.abs(np.linalg.norm(rim.astype(np.float64), axis=2) - 1.0).max() < 2e-7 and (pos.reshape(-1, 3, 3)[:, 0] == 0).all()
    t = pos.reshape(-1, 3, 3).astype(np.float64)
    cr = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
    assert (cr[:, 2] > 0).all()  # counter-clockwise seen from +z: an area emitter on it shines along +z
    assert abs(0.5 * cr[:, 2].sum() - 32.0 * np.sin(2 * np.pi / 64)) < 1e-6
    pos, nrm, uv = ml.builtin_sphere()
    assert pos.shape == (1536, 3) and np.array_equal(pos, nrm)
    assert np.abs(np.linalg.norm(pos.astype(np.float64), axis=1) - 1.0).max() < 2e-7
    t = pos.reshape(-1, 3, 3)
    t64 = t.astype(np.float64)
    assert (np.einsum("ij,ij->i", np.cross(t64[:, 1] - t64[:, 0], t64[:, 2] - t64[:, 0]), t64.mean(1)) > 0).all()  # outward
    edges = {}
    for f in t:
        for a, b in ((0, 1), (1, 2), (2, 0)):
            edges[(f[a].tobytes(), f[b].tobytes())] = edges.get((f[a].tobytes(), f[b].tobytes()), 0) + 1
    assert len(edges) == 1536 and all(n == 1 and edges.get((b, a)) == 1 for (a, b), n in edges.items())  # watertight, bit for bit
    assert abs(0.5 * np.linalg.norm(np.cross(t64[:, 1] - t64[:, 0], t64[:, 2] - t64[:, 0]), axis=1).sum() / (4 * np.pi) - 1.0) < 0.02


def test_cpp_loader_builtin_shapes_on_staircase2(staircase2_xml):
    """The five `<shape type="disk">` ceiling lights of 'Modern Hall', which the reference maps to an assets/disk.obj its
    checkout does not hold (Loader.cpp:276): skipped with a warning by default (the reference's behaviour, the test above),
    BUILT with LoadOptions::builtinShapes -- 5 x 64 triangles and as many triangle lights more, the same arrays from the C++
    loader, the numpy restatement and the fixture the GPU tests use."""
    from gpuspectral_amd import abi, host
    from oracle import mitsuba_loader as ml

    s = host.Scene(staircase2_xml, builtin_shapes=True)
    a, b = s.arrays(), ml.load_scene(staircase2_xml, builtin_shapes=True)
    same_scene(a, b)
    assert a.num_triangles == 30927 + 5 * 64 and len(a.lights) == 16 + 5 * 64 and len(a.instances) == 32
    assert not any("disk" in w for w in s.warnings)
    cached = abi.SceneArrays.load(os.path.join(os.path.dirname(os.path.dirname(CORNELL_XML)), "ref_scenes", "staircase2_shapes.npz"))
    same_scene(a, cached)
    # the disks are ceiling lights: their +z (the side an area emitter on Mitsuba's disk shines from) maps to world -y, and
    # so do the geometric normals of their 320 triangle lights
    disks = a.instances[a.instances["vertex_count"] == 192]
    assert len(disks) == 5 and (disks["transform"][:, 9] < -0.1).all() and (np.abs(disks["transform"][:, [8, 10]]) < 1e-6).all()
    lp = a.lights["positions"][:, :, :3].astype(np.float64)
    n = np.cross(lp[:, 1] - lp[:, 0], lp[:, 2] - lp[:, 0])
    area = 0.5 * np.linalg.norm(n, axis=1)
    small = np.abs(area - 0.5 * np.sin(2 * np.pi / 64) * 0.104916 ** 2) < 1e-7
    assert small.sum() == 320 and (n[small, 1] < 0).all() and (np.abs(n[small][:, [0, 2]]).max(1) < 1e-5 * np.abs(n[small, 1])).all()
    d = host.Scene(staircase2_xml)  # default: skipped, said so
    assert d.arrays().num_triangles == 30927 and sum("disk.obj" in w for w in d.warnings) == 5


def test_cpp_loader_builtin_shapes_on_living_room():
    """'The Modern Living Room' with its sphere light BUILT (radius 0.164 about `center`): 512 triangles and lights more;
    C++ loader == numpy restatement == the GPU tests' fixture.  Needs the reference tree (build container)."""
    from gpuspectral_amd import abi, host
    from oracle import mitsuba_loader as ml

    xml = os.path.join(REF_SCENES, "living-room", "scene.xml")
    if not os.path.exists(xml):
        pytest.skip("reference tree not available")
    s = host.Scene(xml, builtin_shapes=True)
    a, b = s.arrays(), ml.load_scene(xml, builtin_shapes=True)
    same_scene(a, b)
    assert a.num_triangles == 295904 + 512 and len(a.lights) == 512 and len(a.instances) == 29
    cached = abi.SceneArrays.load(os.path.join(os.path.dirname(os.path.dirname(CORNELL_XML)), "ref_scenes", "living-room_shapes.npz"))
    same_scene(a, cached)
    lp = a.lights["positions"][:, :, :3].astype(np.float64)
    c = np.array([-4.50891, 1.81441, -3.77121])
    assert np.abs(np.linalg.norm(lp - c, axis=2) - 0.164157).max() < 1e-5


def test_tone_map_and_ppm(tmp_path):
    """Presentation step: gamma 2.2 (ldrfilm) and the reference's dormant ACES fit (common.glsl:74-82)."""
    from gpuspectral_amd import host

    x = np.zeros((2, 3, 4), np.float32)
    x[0, 0, :3] = (0.0, 0.5, 1.0)
    x[0, 1, :3] = (2.0, -1.0, np.nan)
    x[1, 2, :3] = (0.18, 0.18, 0.18)
    ldr = host.tone_map(x)
    assert ldr[0, 0].tolist() == [0, int(0.5 ** (1 / 2.2) * 255 + 0.5), 255]
    assert ldr[0, 1].tolist() == [255, 0, 0]
    aces = host.tone_map(x, aces=True)
    f = lambda v: min(max((v * (2.51 * v + 0.03)) / (v * (2.43 * v + 0.59) + 0.14), 0.0), 1.0)
    assert abs(int(aces[1, 2, 0]) - int(f(0.18) ** (1 / 2.2) * 255 + 0.5)) <= 1
    p = str(tmp_path / "o.ppm")
    host.write_ppm(p, x)
    raw = open(p, "rb").read()
    assert raw.startswith(b"P6\n3 2\n255\n") and len(raw) == len(b"P6\n3 2\n255\n") + 18


def test_scene_arrays_npz_round_trip(tmp_path, cornell):
    """SceneArrays.save / load (the scene cache scripts/export_reference_scenes.py ships to the GPU box)."""
    from gpuspectral_amd import abi

    f = str(tmp_path / "scene.npz")
    cornell.save(f)
    back = abi.SceneArrays.load(f)
    assert back.num_triangles == cornell.num_triangles
    assert np.array_equal(back.instances, cornell.instances) and np.array_equal(back.positions, cornell.positions)
    assert np.array_equal(back.normals, cornell.normals) and np.array_equal(back.lights, cornell.lights)
    for a, b in zip(back.bsdfs, cornell.bsdfs):
        assert np.array_equal(a, b)
    assert np.array_equal(back.to_world, cornell.to_world) and back.fov == cornell.fov


def test_usable_cpus_respects_cgroup_quota(oracle_mod, monkeypatch, tmp_path):
    """cpu_baseline.cores must be the CPUs actually granted (the GPU boxes show 256 threads, quota 16)."""
    import builtins

    n = oracle_mod.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    real_open = builtins.open

    def fake_open(path, *a, **k):
        if path == "/sys/fs/cgroup/cpu.max":
            p = tmp_path / "cpu.max"
            p.write_text("200000 100000\n")
            return real_open(p, *a, **k)
        return real_open(path, *a, **k)

    monkeypatch.setattr(builtins, "open", fake_open)
    assert oracle_mod.usable_cpus() == min(2, len(os.sched_getaffinity(0)))


def test_cli_rejects_a_malformed_device_list(tmp_path):
    """gsp_render's device list: digits separated by single commas.  Garbage used to render silently on device 0
    (ADVICE r02); it is a usage error now, reported before anything touches a GPU."""
    import subprocess

    lib = os.path.join(ROOT, "gpuspectral_amd", "lib")
    exe = os.path.join(lib, "gsp_render")
    if not os.path.exists(exe):
        pytest.skip("host CLI not built")
    env = dict(os.environ, LD_LIBRARY_PATH=lib + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for bad in ("abc", "0,,1", "-1", "0,1x", "", ",0", "0,"):
        r = subprocess.run([exe, CORNELL_XML, str(tmp_path / "x.pfm"), "8", "8", "1", bad], env=env, capture_output=True, text=True, timeout=60)
        assert r.returncode == 2 and "bad device list" in r.stderr, (bad, r.returncode, r.stderr)
