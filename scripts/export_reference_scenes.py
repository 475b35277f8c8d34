"""Flatten the reference's shipped Mitsuba scenes with the C++ loader (gpuspectral_amd/host) and cache the POD
arrays under scene_cache/ (git-ignored; travels to the GPU box with gpurun).  Runs in the build container, where
/root/reference is mounted; the cache holds loader OUTPUT (vertex arrays, BSDF records), not reference files.
SURVEY 8(f).1: the reference's own scenes as additional benchmarks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpuspectral_amd import host

ROOT = "/root/reference/src/GPUSpectral/assets"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scene_cache")
os.makedirs(OUT, exist_ok=True)
for name in sys.argv[1:] or ["cornell-box", "coffee", "staircase2", "living-room"]:
    sc = host.Scene(os.path.join(ROOT, "scenes", name, "scene.xml"), ROOT)
    a = sc.arrays()
    a.save(os.path.join(OUT, name + ".npz"))
    hist = {n: len(b) for n, b in zip(("diffuse", "dielectric", "conductor", "plastic", "roughconductor", "floor", "roughfloor", "roughplastic"), a.bsdfs) if len(b)}
    print("%-12s %7d tris %4d instances %5d lights  bsdfs %s  warnings %d" % (name, a.num_triangles, len(a.instances), len(a.lights), hist, len(sc.warnings)))
