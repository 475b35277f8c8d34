"""Hand-run: staircase2 with and without the loader's dormant features against the Tungsten fixture, per cell."""
import os
import sys
import tarfile
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import test_gpu_reference_images as T  # noqa: E402
from gpuspectral_amd import host  # noqa: E402

REF = T.REF
d = tempfile.mkdtemp()
for n in ("staircase2.tar.xz", "staircase2_textures.tar"):
    with tarfile.open(os.path.join(REF, n)) as t:
        t.extractall(d)
xml = os.path.join(d, "staircase2", "scene.xml")
fix = np.load(os.path.join(REF, "tungsten_staircase2.npz"))
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
np.set_printoptions(precision=2, linewidth=200, suppress=True)
out = {}
for name, kw in (("reference behaviour", {}), ("dormant features, sRGB", dict(dormant_features=True)),
                 ("dormant features, linear bytes", dict(dormant_features=True, srgb_textures=False))):
    s = host.Scene(xml, **kw)
    sc = s.arrays()
    ours = T._render(sc, 1024, 1024, spp)
    r, ok = T._cells(ours, fix)
    out[name] = r
    lum = np.nanmean(r, axis=2)
    print("==", name, "| textures", len(sc.textures), "| warnings", len(s.warnings))
    print("cell ratio ours / Tungsten (mean over channels):")
    print(lum)
    v = np.isfinite(r)
    print("all cells: n %d median %.3f  mean |log2 ratio| %.3f" % (v.sum(), np.nanmedian(r), np.nanmean(np.abs(np.log2(r[v])))))
    for label, sl in (("right wall rows 0-3 cols 4-7", (slice(0, 4), slice(4, 8))), ("lower half rows 4-7", (slice(4, 8), slice(0, 8))),
                      ("left half cols 0-3", (slice(0, 8), slice(0, 4)))):
        q = r[sl]
        qv = np.isfinite(q)
        print("   %-30s n %3d  min %.2f max %.2f median %.2f  mean|log2| %.3f  chroma err %.3f" % (
            label, qv.sum(), np.nanmin(q), np.nanmax(q), np.nanmedian(q), np.nanmean(np.abs(np.log2(q[qv]))),
            np.nanmean(np.abs(q / np.nanmean(q, axis=2, keepdims=True) - 1.0))))
