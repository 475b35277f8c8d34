#!/usr/bin/env python3
"""The reference's own shader text, executed: tests/golden/glsl_vectors.npz.

    python tests/golden/make_glsl_vectors.py              # build oracle/_ref/libglsl_ref.so, write the fixture
    python tests/golden/make_glsl_vectors.py --build-only # only the library (what __graft_entry__.build() runs)
    python tests/golden/make_glsl_vectors.py --check      # regenerate in memory, compare with the committed fixture

Needs /root/reference (the build container); nothing here runs on the GPU box -- only the .npz travels.

What it does.  The pure functions of the reference's shaders -- RNG, Onb, the samplers, the Fresnel terms, the eight BSDF
sample / eval pairs, their dispatch and the light sampling (S/assets/shaders/pt_common.glsl:30-42,86-151 and
rayhit.rchit:17-70,89-654) -- are read from the reference tree where it lies, given three token-level rewrites
    out T x          ->  T& x        (GLSL out parameter -> C++ reference)
    1.0, 0.5, 0.01   ->  1.0f, ...   (an unsuffixed real literal IS a 32-bit float in GLSL; in C++ it would be a double)
    .xyz             ->  .xyz()      (the one swizzle the text uses, rayhit.rchit:132-134)
and compiled as C++ (gcc, -ffp-contract=off) against oracle/glsl_shim.h, which supplies vec2 / vec3 / vec4, the GLSL
built-ins (mapped to the ONE implementation oracle_math.h fixes for them) and left-to-right argument evaluation.  The
rewritten text exists only in a temporary directory for the duration of the compile; the product of the build is
oracle/_ref/libglsl_ref.so (git-ignored).  oracle/glsl_harness.cpp then calls sampleBSDF / evalBSDF / sampleLight / onb* /
tea / pcgHash / randPcg / randUniform on seeded inputs and this script stores inputs and outputs.

Second step (r06): the shaders' main() functions as well -- raygen.rgen:20-108 (rayDir, the bounce loop, cutoff, Russian roulette,
running mean), rayhit.rchit:656-797 (NEE, isvalid, the closest-hit main), miss.rmiss:15-18, shadowmiss.rmiss:6-9, with
pt_common.glsl:4-28,44-51 (HitPayload, Camera, RenderParams, Instance) -- compiled the same way (one more rewrite: `.xy` -> `.xy()`)
and run for whole frames (IMAGES below).  What the ray-tracing pipeline supplies is supplied by oracle/glsl_harness.cpp: the gl_*
built-in variables, the payloads, the storage image, and traceRayEXT = the DRIVER's traversal (vendor-opaque; the oracle's BVH and
triangle test stand in for it through a callback into liboracle_pt.so) followed by the shader the hit or miss selects.  The frames
go into the fixture as img_<name> (+ the extension / shadow ray counts).

What it is worth.  It is NOT a run of the reference (no glslc, no Vulkan here, and the built-ins' arithmetic is the shim's
choice), so by the grading rule it pins nothing and parity stays "partial".  What it replaces is the human READING: every
operator, every grouping, every branch and every literal of 566 lines of shader now reaches the fixture through a
compiler, not through somebody's restatement.  tests/test_glsl_vectors.py holds the oracle (oracle/oracle_bsdf.h, oracle_pt.cpp)
and the product's own headers (gpuspectral_amd/csrc/pt_shading.h, pt_stages.h via tests/emu) to these vectors and frames bit for
bit; tests/test_gpu_parity.py::test_gpu_frames_equal_the_executed_reference_shaders does the same for the HIP kernels.
"""
import argparse
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
SHADERS = "/root/reference/src/GPUSpectral/assets/shaders"
OUT_SO = os.path.join(ROOT, "oracle", "_ref", "libglsl_ref.so")
FIXTURE = os.path.join(HERE, "glsl_vectors.npz")

# (macro, file, first line, last line, text the first line must start with, text the last line must start with)
PARTS = [
    ("GLSL_PART_PAYLOAD", "pt_common.glsl", 4, 28, "struct HitPayload {", "};"),
    ("GLSL_PART_INSTANCE", "pt_common.glsl", 44, 51, "struct Instance {", "};"),
    ("GLSL_PART_RCHIT_MAIN", "rayhit.rchit", 656, 798, "#define NEE true", "}"),
    ("GLSL_PART_MISS", "miss.rmiss", 15, 18, "void main()", "}"),
    ("GLSL_PART_SHADOWMISS", "shadowmiss.rmiss", 6, 9, "void main()", "}"),
    ("GLSL_PART_RGEN", "raygen.rgen", 20, 109, "vec3 rayDir(", "}"),
    ("GLSL_PART_HANDLES", "pt_common.glsl", 30, 42, "#define BSDFHandle uint", "}"),
    ("GLSL_PART_RNG_ONB", "pt_common.glsl", 86, 151, "uint rngState;", "}"),
    ("GLSL_PART_STRUCTS", "rayhit.rchit", 17, 70, "struct TriangleLight {", "};"),
    ("GLSL_PART_FUNCTIONS", "rayhit.rchit", 89, 654, "vec2 sampleConcentric() {", "}"),
]

REAL = re.compile(r"(?<![\w.])(\d+\.\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)(?![\w.])")


def rewrite(text):
    text = re.sub(r"\bout\s+(\w+)\s+(\w+)", r"\1& \2", text)
    text = REAL.sub(lambda m: m.group(1) + "f", text)
    text = re.sub(r"\.xyz\b", ".xyz()", text)
    return re.sub(r"\.xy\b", ".xy()", text)  # (gl_LaunchIDEXT.xy, raygen.rgen:32,85,107)


def build():
    if not os.path.isdir(SHADERS):
        raise SystemExit("make_glsl_vectors: %s is not mounted (this script runs in the build container only)" % SHADERS)
    os.makedirs(os.path.dirname(OUT_SO), exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        defs = []
        for macro, name, a, b, first, last in PARTS:
            lines = open(os.path.join(SHADERS, name), encoding="utf-8", errors="replace").read().split("\n")
            part = lines[a - 1:b]
            assert part[0].strip().startswith(first) and part[-1].strip().startswith(last), (name, a, b, part[0], part[-1])
            path = os.path.join(tmp, macro + ".inc")
            with open(path, "w") as fh:
                fh.write(rewrite("\n".join(part)) + "\n")
            defs.append('-D%s="%s"' % (macro, path))
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-Wall", "-Wno-unused-function",
               "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-shared", "-I", os.path.join(ROOT, "oracle"), "-I", SHADERS] + defs + [
                   os.path.join(ROOT, "oracle", "glsl_harness.cpp"), "-o", OUT_SO]
        subprocess.check_call(cmd)
    return OUT_SO


def tables():
    """The BSDF records and lights the vectors run on: four records per type, ordinary and edge parameters."""
    from gpuspectral_amd import abi

    s = abi.SceneArrays()
    B = [np.zeros(4, dt) for dt in abi.BSDF_DTYPES]
    B[0]["reflectance"] = [(0.725, 0.71, 0.68), (0.14, 0.45, 0.091), (0.0, 0.0, 0.0), (1.0, 0.5, 2.0)]
    B[1]["ior_in"], B[1]["ior_out"] = [1.5, 1.33, 1.0, 2.4], [1.0, 1.0, 1.5, 1.000277]
    B[2]["ior_in"], B[2]["ior_out"] = [0.0, 1.5, 0.2, 3.0], [1.0, 1.0, 1.0, 1.5]
    B[3]["diffuse"] = [(0.5, 0.5, 0.5), (0.9, 0.1, 0.1), (0.0, 0.0, 0.0), (0.2, 0.6, 0.95)]
    B[3]["ior_in"], B[3]["ior_out"], B[3]["r0"] = [1.5, 1.9, 1.49, 1.1], [1.0, 1.0, 1.000277, 1.0], [0.04, 0.0963, 0.0387, 0.00227]
    B[4]["eta"] = [(0.2004, 0.924, 1.102), (0.143, 0.375, 1.442), (1.657, 0.880, 0.521), (4.37, 2.92, 2.35)]
    B[4]["k"] = [(3.912, 2.452, 2.142), (3.983, 2.386, 1.603), (9.224, 6.27, 4.837), (3.3, 3.0, 2.5)]
    B[4]["reflectance"] = [(1.0, 1.0, 1.0), (0.9, 0.8, 0.7), (1.0, 1.0, 1.0), (0.5, 0.5, 0.5)]
    B[4]["alpha"] = [0.1, 0.01, 0.3, 0.8]
    B[5]["diffuse"], B[5]["r0"] = [(0.4, 0.4, 0.4), (0.8, 0.7, 0.2), (0.05, 0.05, 0.05), (1.0, 1.0, 1.0)], [0.04, 0.1, 0.5, 0.0]
    B[6]["diffuse"] = [(0.4, 0.4, 0.4), (0.8, 0.7, 0.2), (0.05, 0.05, 0.05), (1.0, 1.0, 1.0)]
    B[6]["r0"], B[6]["alpha"] = [0.04, 0.1, 0.5, 0.02], [0.05, 0.2, 0.5, 1.0]
    B[7]["diffuse"] = [(0.5, 0.5, 0.5), (0.9, 0.1, 0.1), (0.0, 0.0, 0.0), (0.2, 0.6, 0.95)]
    B[7]["ior_in"], B[7]["ior_out"], B[7]["r0"] = [1.5, 1.9, 1.49, 1.1], [1.0, 1.0, 1.000277, 1.0], [0.04, 0.0963, 0.0387, 0.00227]
    B[7]["alpha"] = [0.1, 0.01, 0.3, 0.7]
    s.bsdfs = B
    L = np.zeros(6, abi.LIGHT_DT)
    tri = [((-0.24, 1.98, -0.22), (0.23, 1.98, -0.22), (0.23, 1.98, 0.16)),      # the Cornell light's two triangles
           ((-0.24, 1.98, -0.22), (0.23, 1.98, 0.16), (-0.24, 1.98, 0.16)),
           ((1.0, 0.2, 0.3), (1.0, 1.4, 0.1), (1.0, 0.9, -0.8)),                   # a wall light
           ((0.0, 0.0, 0.0), (1e-3, 0.0, 0.0), (0.0, 1e-3, 0.0)),                  # tiny
           ((-3.0, 4.0, 2.0), (5.0, 4.5, -1.0), (0.5, 3.0, 6.0)),                  # large, tilted
           ((0.3, 0.3, 0.3), (0.6, 0.6, 0.6), (0.9, 0.9, 0.9))]                    # degenerate (zero area): NaN / inf must agree too
    for i, t in enumerate(tri):
        L["positions"][i, :, :3] = t
        L["positions"][i, :, 3] = 1.0
    L["radiance"][:, :3] = [(17, 12, 4), (17, 12, 4), (5, 5, 5), (100, 0, 0), (0.5, 0.7, 0.9), (1, 1, 1)]
    s.lights = L
    # a geometry the oracle / the emulator can build a tree over (the vectors never trace)
    s.positions = np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0)], np.float32)
    s.normals = np.array([(0, 0, 1)] * 3, np.float32)
    inst = np.zeros(1, abi.INSTANCE_DT)
    inst["transform"][0] = np.eye(4, dtype=np.float32).reshape(16)
    inst["vertex_count"][0] = 3
    inst["bsdf"][0] = abi.bsdf_handle(0, 0)
    s.instances = inst
    return s


def unit(rng, n):
    v = rng.normal(size=(n, 3))
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def inputs():
    from gpuspectral_amd import abi

    rng = np.random.RandomState(20261004)
    per = 400
    handles, wo = [], []
    special = np.array([(0, 0, 1), (0, 0, -1), (1, 0, 0), (0, 1, 0), (0.6, 0.8, 0), (1e-4, 0, 1), (0.70710678, 0, 0.70710678),
                        (1, 0, 1e-4), (1, 0, -1e-4), (1e-20, 0, 1e-20), (0.5, 0.5, 0.70710678), (-0.3, 0.2, 0.93273791)], np.float32)
    for t in range(8):
        for i in range(4):
            w = unit(rng, per)
            up = rng.uniform(size=per) < 0.7  # most directions arrive from the outside
            w[up, 2] = np.abs(w[up, 2])
            w[:len(special)] = special
            handles += [abi.bsdf_handle(t, i)] * per
            wo.append(w)
    handles = np.array(handles, np.uint32)
    wo = np.concatenate(wo)
    seeds = rng.randint(0, 2**32, size=len(handles), dtype=np.uint64).astype(np.uint32)
    nl = 3000
    lpos = (rng.uniform(-1, 1, (nl, 3)) * [1.2, 1.0, 1.2] + [0, 1, 0]).astype(np.float32)
    lseeds = rng.randint(0, 2**32, size=nl, dtype=np.uint64).astype(np.uint32)
    no = 2000
    onb_n = (unit(rng, no) * rng.uniform(0.1, 3.0, (no, 1))).astype(np.float32)
    onb_n[:6] = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (-1, 0, 0), (0, -1, 0), (0, 0, -1)]
    onb_n[6:8] = [(0.70710678, 0, 0.70710678), (0.70710678, 0, -0.70710678)]  # |x| == |z|: the branch of pt_common.glsl:131
    onb_v = unit(rng, no)
    rng_a = rng.randint(0, 2**32, size=1000, dtype=np.uint64).astype(np.uint32)
    rng_b = rng.randint(0, 2**32, size=1000, dtype=np.uint64).astype(np.uint32)
    rng_a[:4], rng_b[:4] = [0, 1, 0xFFFFFFFF, 128 * 9 + 5], [0, 0, 0xFFFFFFFF, 3]
    pf = np.concatenate([rng.uniform(0, 50, 500), [0, 0, 1e-30, 1e30, np.inf]]).astype(np.float32)
    pg = np.concatenate([rng.uniform(0, 50, 500), [0, 1, 1e-30, 1e30, 1]]).astype(np.float32)
    return dict(handles=handles, wo=wo, seeds=seeds, light_pos=lpos, light_seeds=lseeds, onb_n=onb_n, onb_v=onb_v, rng_a=rng_a,
                rng_b=rng_b, helper_f=pf, helper_g=pg), rng


def run(lib):
    L = C.CDLL(lib)
    sc = tables()
    desc = sc.desc()
    L.glsl_set_tables(C.byref(desc))
    inp, rng = inputs()
    n = len(inp["handles"])
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    samples = np.zeros((n, 9), np.uint32)
    L.glsl_bsdf_sample(C.c_uint64(n), p(inp["handles"]), p(inp["wo"]), p(inp["seeds"]), p(samples))
    # eval directions: the sampled wi itself (what MIS evaluates) for half the vectors, a free direction for the rest
    wi = unit(rng, n)
    own = rng.uniform(size=n) < 0.5
    swi = samples[:, :3].copy().view(np.float32)
    own &= np.isfinite(swi).all(axis=1)
    wi[own] = swi[own]
    wi = np.ascontiguousarray(wi)
    evals = np.zeros((n, 5), np.uint32)
    L.glsl_bsdf_eval(C.c_uint64(n), p(inp["handles"]), p(inp["wo"]), p(wi), p(evals))
    nl = len(inp["light_pos"])
    lights = np.zeros((nl, 8), np.uint32)
    L.glsl_sample_light(C.c_uint64(nl), p(inp["light_pos"]), p(inp["light_seeds"]), p(lights))
    no = len(inp["onb_n"])
    onb = np.zeros((no, 15), np.uint32)
    L.glsl_onb(C.c_uint64(no), p(inp["onb_n"]), p(inp["onb_v"]), p(onb))
    nr = len(inp["rng_a"])
    rngo = np.zeros((nr, 6), np.uint32)
    L.glsl_rng(C.c_uint64(nr), p(inp["rng_a"]), p(inp["rng_b"]), p(rngo))
    nh = len(inp["helper_f"])
    hh = inp["handles"][:: max(1, n // nh)][:nh].copy()
    helpers = np.zeros((nh, 3), np.uint32)
    L.glsl_helpers(C.c_uint64(nh), p(inp["helper_f"]), p(inp["helper_g"]), p(hh), p(helpers))
    out = dict(inp)
    out.update(wi=wi, samples=samples, evals=evals, lights=lights, onb=onb, rng=rngo, helper_handles=hh, helpers=helpers)
    out.update(images(L))
    for i, b in enumerate(sc.bsdfs):
        out["bsdf%d" % i] = b
    out["light_table"] = sc.lights
    return out


# whole frames through the shaders' main() functions (raygen.rgen:29-108, rayhit.rchit:666-797, the two miss shaders): name ->
# (scene, width, height, spp).  The driver's traversal is the oracle's (oracle_trace); everything else is the reference's text.
IMAGES = {"cornell": (96, 96, 8), "materials": (96, 80, 4)}
# ... and random scenes of the parity fuzzer (tests/tools/fuzz_parity.py::random_scene: every BSDF type at ordinary and EXTREME
# parameters -- alpha 1e-4, ior 1, zero / > 1 reflectance, k = 50 --, mirrored and sheared instance transforms, several lights, random
# cameras): the corners where NaN / inf handling and the termination tests of rayhit.rchit:770-784 decide the image
IMAGES.update({"fuzz%d" % seed: (40, 28, 3) for seed in range(1100000, 1100024)})


def image_scene(name):
    from gpuspectral_amd import scenes
    from oracle import mitsuba_loader as ml

    if name == "cornell":
        return ml.load_scene(os.path.join(HERE, "cornell-box", "scene.xml"))
    if name.startswith("fuzz"):
        sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
        import fuzz_parity

        return fuzz_parity.random_scene(int(name[4:]))
    return scenes.cornell_materials(8)  # all eight BSDF types, glass (deep delta paths), a mirror


def images(L):
    from oracle import oracle as orc

    OL = orc.lib()
    out = {}
    for name, (w, h, spp) in IMAGES.items():
        sc = image_scene(name)
        o = orc.Oracle(sc)
        invt = np.concatenate([orc.transform_inv_t(inst["transform"]) for inst in sc.instances]).astype(np.float32)
        desc = sc.desc()
        L.glsl_set_scene(C.byref(desc), invt.ctypes.data_as(C.c_void_p), C.cast(OL.oracle_trace, C.c_void_p), C.c_void_p(o._h))
        accum = np.zeros((h, w, 4), np.float32)
        rays = np.zeros(2, np.uint64)
        L.glsl_render(C.c_uint32(w), C.c_uint32(h), C.c_uint32(spp), C.c_uint32(0), accum.ctypes.data_as(C.c_void_p), rays.ctypes.data_as(C.c_void_p))
        out["img_" + name] = accum
        out["img_" + name + "_rays"] = rays
        o.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    lib = build()
    print("built", lib)
    if a.build_only:
        return
    out = run(lib)
    if a.check:
        old = np.load(FIXTURE)
        bad = [k for k in out if not (k in old and np.array_equal(np.asarray(out[k]).view(np.uint8), old[k].view(np.uint8)))]
        print("check against", FIXTURE, "->", "identical" if not bad else "DIFFERENT: %s" % bad)
        raise SystemExit(1 if bad else 0)
    np.savez_compressed(FIXTURE, **out)
    print(FIXTURE, "%d BSDF sample + eval pairs, %d light samples, %d frames, %d RNG rows, %.0f kB"
          % (len(out["handles"]), len(out["lights"]), len(out["onb"]), len(out["rng"]), os.path.getsize(FIXTURE) / 1e3))


if __name__ == "__main__":
    main()
