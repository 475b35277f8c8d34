"""How many host cores does this box really give us?  (cpu_baseline.cores must be what was actually used.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket|MHz' | head -8")
from gpuspectral_amd import scenes
from oracle import oracle as orc
sc = scenes.interior(200_000, seed=7)
o = orc.Oracle(sc)
ids = scenes.tile_pixel_ids(1920, 1080, 0, 16, tile=16)
for nt in (1, 2, 4, 8, 16, 32, 64, 128, 256):
    _, st = o.render(1920, 1080, spp=1, pixel_ids=ids, threads=nt)
    r = st["extension_rays"] + st["shadow_rays"]
    print("%3d threads: %.3f s  %.2f Mrays/s  %.0f krays/s/thread" % (nt, st["seconds"], r / st["seconds"] / 1e6, r / st["seconds"] / 1e3 / nt), flush=True)
