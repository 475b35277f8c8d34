"""How much would ORDERING the bounce rays buy the closest-hit kernel?  4 M secondary rays of the bench scene (origins = hit
points of camera rays, directions uniform on the sphere) through gsp_trace in four orders: shuffled, as the pixels generated them,
grouped by direction octant, octant + Morton code of the origin.  Kernel times by rocprofv3:
   rocprofv3 --kernel-trace --stats -d out -- python3 scripts/experiments/r05_ray_order_probe.py
(the four k_trace<TestIO> launches appear in this order; the script prints hits per order as a cross-check)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpuspectral_amd as g
from gpuspectral_amd import scenes

sc = scenes.interior(1_000_000, seed=7)
rng = np.random.RandomState(1)
W, H = 2048, 2048
with g.Context(0) as ctx:
    ctx.upload_scene(sc)
    m = np.array(sc.to_world, np.float32).reshape(4, 4)
    eye = m[3, :3]
    ys, xs = np.mgrid[0:H, 0:W]
    d = np.stack([(xs - W / 2) / W, (ys - H / 2) / W, np.full(xs.shape, 0.8)], -1).reshape(-1, 3).astype(np.float32)
    d = d @ m[:3, :3]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((W * H, 8), np.float32)
    rays[:, :3], rays[:, 3], rays[:, 4:7], rays[:, 7] = eye, 0.0, d, 1e10
    h = ctx.trace(rays)
    ok = h["prim"] >= 0
    t = h["t"] if "t" in h.dtype.names else h[h.dtype.names[0]]
    org = (eye + d * t[:, None] * 0.999)[ok]
    n = len(org)
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    sec = np.zeros((n, 8), np.float32)
    sec[:, :3], sec[:, 3], sec[:, 4:7], sec[:, 7] = org, 1e-4, dirs, 1e10
    octant = (dirs[:, 0] < 0) * 1 + (dirs[:, 1] < 0) * 2 + (dirs[:, 2] < 0) * 4
    lo, hi = org.min(0), org.max(0)
    q = np.minimum(((org - lo) / (hi - lo + 1e-9) * 1024).astype(np.uint64), 1023)

    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249

    morton = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    orders = {"shuffled": rng.permutation(n), "pixel order": np.arange(n), "by octant": np.argsort(octant, kind="stable"),
              "octant + Morton": np.lexsort((morton, octant))}
    for name, idx in orders.items():
        hh = ctx.trace(sec[idx])
        print("%-16s %d rays, %d hits" % (name, n, int((hh["prim"] >= 0).sum())), flush=True)
