"""AddressSanitizer / UBSan / LeakSanitizer runs of the CPU-side native code (the GPU pool has no sanitizer support)."""
import os
import subprocess

import pytest

from conftest import CORNELL_XML, ROOT

REF_ASSETS = "/root/reference/src/GPUSpectral/assets"


def test_cpp_loader_under_asan_ubsan(tmp_path, staircase2_xml):
    host = os.path.join(ROOT, "gpuspectral_amd", "host")
    exe = str(tmp_path / "asan_loader")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined", "-I", host, "-o", exe,
                           os.path.join(ROOT, "tests", "emu", "asan_loader.cpp"), os.path.join(host, "Loader.cpp"),
                           os.path.join(host, "Image.cpp")])
    scenes = [CORNELL_XML, staircase2_xml]
    assets = os.path.dirname(os.path.dirname(CORNELL_XML))
    if os.path.isdir(REF_ASSETS):  # the reference's large scenes, where the tree is mounted
        assets = REF_ASSETS
        scenes = [os.path.join(REF_ASSETS, "scenes", n, "scene.xml") for n in ("cornell-box", "coffee", "staircase2", "living-room")]
    # image readers of the dormant features: intact files + 400 damaged copies of each
    images = []
    try:
        import numpy as np
        from PIL import Image

        import test_images

        pat = test_images.pattern(45, 61)
        for name, kw in (("a.png", {}), ("b.jpg", dict(subsampling=2, quality=80)), ("c.jpg", dict(subsampling=0, quality=90, restart_marker_blocks=2))):
            Image.fromarray(pat).save(str(tmp_path / name), **kw)
            images.append(str(tmp_path / name))
        Image.fromarray(pat).quantize(8).save(str(tmp_path / "p.png"))
        test_images.write_rgbe(str(tmp_path / "e.hdr"), np.random.RandomState(1).randint(90, 160, (6, 40, 4)).astype(np.uint8), True)
        with open(str(tmp_path / "f.pfm"), "wb") as f:
            f.write(b"PF\n5 4\n-1.0\n" + np.arange(60, dtype="<f4").tobytes())
        images += [str(tmp_path / "p.png"), str(tmp_path / "e.hdr"), str(tmp_path / "f.pfm")]
    except ImportError:
        pass
    r = subprocess.run([exe, assets] + scenes + (["--images"] + images if images else []), cwd=str(tmp_path), capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "loader asan ok" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr
    assert not images or "image fuzz:" in r.stdout


def test_oracle_under_asan_ubsan(tmp_path):
    """The oracle's sanitizer build renders a scene with every BSDF type and traces rays (it is the checker of every
    parity test, so it is checked itself)."""
    so = str(tmp_path / "liboracle_pt_asan.so")
    o = os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-mavx2", "-pthread",
                           "-fsanitize=address,undefined", "-shared", "-o", so, os.path.join(o, "oracle_pt.cpp")])
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from gpuspectral_amd import scenes\n"
        "import oracle as O\n"
        "o = O.Oracle(scenes.cornell_materials(8))\n"
        "o.render(40, 30, spp=2)\n"
        "r = np.zeros((500, 8), np.float32); r[:, 1] = 1; r[:, 2] = -3; r[:, 6] = 1; r[:, 7] = 1e10\n"
        "o.trace(r); o.trace(r, any_hit=True)\n"
        "print('oracle asan ok')\n" % (ROOT, o)
    )
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    r = subprocess.run(["python", "-c", code], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ORACLE_LIB_PATH=so, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and "oracle asan ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
