"""INTEGRATION.md path B -- the `PathTracerHip` binding a maintainer adds to the reference tree
(gpuspectral_amd/host/integration/PathTracerHip.h) -- compiled and run.  The reference's own headers are absent from this
environment (18 empty submodules), so the translation unit that includes the binding (tests/emu/stub_main.cpp) supplies a
FrameGraph type, the reference-shaped RenderPassCreator (S/renderer/Renderer.h:22-25) and this repository's mirror of `Scene`:
the binding itself includes nothing but the C header."""
import os
import subprocess

import numpy as np
import pytest

from conftest import CORNELL_XML, ROOT

LIB = os.path.join(ROOT, "gpuspectral_amd", "lib")


def build(tmp_path):
    exe = str(tmp_path / "stub_main")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "emu", "stub_main.cpp"), "-o", exe, "-L", LIB, "-lgpuspectral_host", "-lgpuspectral_pt",
                           "-Wl,-rpath," + LIB])
    return exe


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline() == b"PF\n"
        w, h = map(int, f.readline().split())
        f.readline()
        return np.frombuffer(f.read(), np.float32).reshape(h, w, 3)[::-1]


def test_binding_compiles_against_the_reference_shaped_interfaces(tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe], capture_output=True)
    assert r.returncode == 2  # usage
    txt = open(os.path.join(ROOT, "gpuspectral_amd", "host", "integration", "PathTracerHip.h")).read()
    assert '#include "' not in txt  # nothing of this repository but <gpuspectral_pt.h>: the including TU supplies Scene


@pytest.mark.gpu
def test_binding_renders_and_follows_per_frame_edits(tmp_path, oracle_mod):
    """createRenderPass(fg, scene) per frame through the binding: six frames == the oracle's six samples; with an edit after
    frame 3 (camera, a transform, a BSDF record -- re-read every frame as in PathTracer.cpp:58-93) == the oracle continued on
    the edited scene."""
    from gpuspectral_amd import host

    exe = build(tmp_path)
    W, H = 64, 48
    out = str(tmp_path / "a.pfm")
    r = subprocess.run([exe, CORNELL_XML, out, str(W), str(H), "6"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    sc = host.Scene(CORNELL_XML).arrays()
    ref, _ = oracle_mod.Oracle(sc).render(W, H, spp=6)
    assert np.array_equal(read_pfm(out).reshape(-1, 3), ref[:, :3])
    r = subprocess.run([exe, CORNELL_XML, out, str(W), str(H), "6", "0.25"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    acc, _ = oracle_mod.Oracle(sc).render(W, H, spp=3)
    sc.to_world = np.array(sc.to_world, np.float32).copy()
    sc.to_world[12] += np.float32(0.25)
    inst = sc.instances.copy()
    t = inst["transform"][5].copy()
    t[13] += np.float32(0.25)
    inst["transform"][5] = t
    sc.instances = inst
    b = [x.copy() for x in sc.bsdfs]
    b[0]["reflectance"][0][2] = np.float32(0.9)
    sc.bsdfs = b
    acc, _ = oracle_mod.Oracle(sc).render(W, H, spp=3, first_timestamp=3, accum=acc)
    img = read_pfm(out).reshape(-1, 3)
    assert np.array_equal(img, acc[:, :3])
    assert not np.array_equal(img, ref[:, :3])
