"""The version-ring bookkeeping of the per-frame scene edits (gpuspectral_amd/csrc/pt_versions.h -- the structs pt_render.hip asks
which ring slot to write, whether an edit must wait, how large a ring may be), driven WITHOUT a GPU: tests/emu/versions_model.cpp
runs random streams of render / update_tables / update_instances / drain / upload against a mock device memory, under
AddressSanitizer + UBSan.  Invariants: no slot is written while a sample in flight names it, every sample reads the version it
was generated under (through collapses, lazy ring growth, base changes, wrap-around), the byte ledger never underflows, ring
plans respect their limits, the split limits are what include/gpuspectral_pt.h documents.  (r05 review item 6: this logic was
testable only on hardware.)  The same header is compiled into libgpuspectral_pt.so; tests/test_gpu_scene_updates.py exercises
it on the GPU against the oracle."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "emu", "versions_model.cpp")
HDR = os.path.join(ROOT, "gpuspectral_amd", "csrc", "pt_versions.h")


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("versions") / "versions_model")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra", "-Werror",
                           SRC, "-o", exe])
    return exe


@pytest.mark.parametrize("seed", [1, 2, 3, 20261004])
def test_random_edit_streams(model, seed):
    r = subprocess.run([model, str(seed), "40000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"(\d+) in-place table edits, (\d+) ring growths, (\d+) collapses, (\d+) geometry edits in the ring, (\d+) drains", r.stdout)
    assert m and all(int(x) > 500 for x in m.groups()), r.stdout  # every path of the state machine was walked, many times


def test_the_model_catches_an_off_by_one(tmp_path):
    """The harness is only worth something if a wrong ring test fails it: allow one version too many in flight."""
    for pattern, repl in ((r"ver \+ 1 - oldest_live < slots\(\)", "ver + 1 - oldest_live <= slots()"),
                          (r"return ver \+ 1 - oldest_live < slots \?", "return ver + 1 - oldest_live <= slots ?")):
        text = open(HDR).read()
        assert len(re.findall(pattern, text)) == 1
        hdr = tmp_path / "pt_versions_mut.h"
        hdr.write_text(re.sub(pattern, repl, text))
        src = tmp_path / "vm.cpp"
        src.write_text(open(SRC).read().replace("../../gpuspectral_amd/csrc/pt_versions.h", str(hdr)))
        exe = str(tmp_path / "vm")
        subprocess.check_call(["g++", "-O1", "-std=c++17", str(src), "-o", exe])
        r = subprocess.run([exe, "1", "40000"], capture_output=True, text=True)
        # (the ring's slot test fails its own known-answer check first, the tables' the random stream's invariant)
        assert r.returncode != 0 and ("written while a batch" in r.stderr or "needs_no_wait" in r.stderr), (pattern, r.stdout, r.stderr)


def test_product_and_documentation_agree_on_the_limits():
    """include/gpuspectral_pt.h, DESIGN.md and the code name ONE split limit and ONE ring width."""
    hdr = open(HDR).read()
    splits = int(re.search(r"kMaxSceneSplits = (\d+);", hdr).group(1))
    api = open(os.path.join(ROOT, "include", "gpuspectral_pt.h")).read()
    assert "kMaxSceneSplits = %d" % splits in api or "%d times" % splits in api, "the ABI header must quote the split limit"
    csrc = os.path.join(ROOT, "gpuspectral_amd", "csrc")
    render = "".join(open(os.path.join(csrc, f)).read() for f in ("pt_render.hip", "pt_render_kernels.inc", "pt_render_scene.inc", "pt_render_pipeline.inc"))
    assert "scene_splits >= kMaxSceneSplits" in render and not re.search(r"scene_splits >= \d", render)
    stages = open(os.path.join(ROOT, "gpuspectral_amd", "csrc", "pt_stages.h")).read()
    assert int(re.search(r"kTableVersions = (\d+)", stages).group(1)) == int(re.search(r"kMaxTableVersions = (\d+);", hdr).group(1))
    assert int(re.search(r"kGeoVersions = (\d+)", stages).group(1)) == int(re.search(r"kMaxGeoVersions = (\d+);", hdr).group(1))
