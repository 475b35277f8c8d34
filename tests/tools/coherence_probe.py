"""Hand-run under `rocprofv3 --kernel-trace --stats`: how much faster does the closest-hit traversal get when the rays
of a launch are ordered for coherence?  Secondary rays of the bench scene (diffuse bounces 1..3 from the camera's first
hits, built with gsp_trace itself) are traced in four orders; every order is its own k_trace<..TestIO> launch, in this
sequence (2 warm-up launches first):  queue order (as the bounce produced them: by pixel), random permutation,
sorted by direction octant then origin Morton code, sorted by origin Morton code then direction octant.
The per-launch durations come from the kernel trace (the call itself is dominated by the PCIe copies)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import gpuspectral_amd as g  # noqa: E402
from gpuspectral_amd import scenes  # noqa: E402


def morton3(q):  # q: (n, 3) uint32 < 1024
    def spread(v):
        v = v.astype(np.uint64) & 0x3FF
        v = (v | (v << 16)) & 0x30000FF
        v = (v | (v << 8)) & 0x300F00F
        v = (v | (v << 4)) & 0x30C30C3
        v = (v | (v << 2)) & 0x9249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def cosine_dirs(rng, n_vec):
    n = len(n_vec)
    u1, u2 = rng.rand(n), rng.rand(n)
    r, phi = np.sqrt(u1), 2 * np.pi * u2
    l = np.stack([r * np.cos(phi), r * np.sin(phi), np.sqrt(1 - u1)], 1)
    a = np.where(np.abs(n_vec[:, :1]) > 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    t = np.cross(a, n_vec)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(n_vec, t)
    return (l[:, :1] * t + l[:, 1:2] * b + l[:, 2:] * n_vec).astype(np.float32)


sc = scenes.interior(1_000_000)
W, H = 1920, 1080
rng = np.random.RandomState(0)
with g.Context(0) as ctx:
    ctx.upload_scene(sc)
    # camera rays through the C ABI's own frame: reuse the oracle-free route -- positions from a 1-spp render are not
    # exposed, so build primary rays from the camera matrix like raygen.rgen does
    tw = np.array(sc.to_world, np.float32).reshape(4, 4).T  # column-major memory -> matrix
    py, px = np.mgrid[0:H, 0:W]
    z = (max(W, H) / 2.0) / np.tan(float(sc.fov) / 2.0)
    dl = np.stack([-(px - W / 2.0), (py - H / 2.0), np.full(px.shape, z)], -1).reshape(-1, 3)
    dl /= np.linalg.norm(dl, axis=1, keepdims=True)
    d = dl @ tw[:3, :3].T
    d[:, 1] *= -1.0
    o = np.broadcast_to(tw[:3, 3], d.shape)
    rays = np.zeros((W * H, 8), np.float32)
    rays[:, 0:3], rays[:, 4:7], rays[:, 7] = o, d, 1e10
    lo, hi = sc.positions.min(0) - 5, sc.positions.max(0) + 5
    for bounce in range(1, 4):
        hits = ctx.trace(rays)
        ok = hits["prim"] >= 0
        p = rays[ok, 0:3] + rays[ok, 4:7] * hits["t"][ok, None]
        # shading normal unknown here: bounce about the reversed ray direction's hemisphere (a diffuse-like spread)
        nrm = -rays[ok, 4:7]
        nd = cosine_dirs(rng, nrm.astype(np.float64))
        rays = np.zeros((ok.sum(), 8), np.float32)
        rays[:, 0:3], rays[:, 4:7], rays[:, 7] = p + 1e-3 * nd, nd, 1e10
        n = len(rays)
        octant = ((rays[:, 4] < 0).astype(np.uint64) | ((rays[:, 5] < 0).astype(np.uint64) << 1) | ((rays[:, 6] < 0).astype(np.uint64) << 2))
        # scene bounds for the Morton grid: from the ray origins themselves
        olo, ohi = rays[:, 0:3].min(0), rays[:, 0:3].max(0)
        q = np.clip((rays[:, 0:3] - olo) / np.maximum(ohi - olo, 1e-6) * 1023.0, 0, 1023).astype(np.uint32)
        m = morton3(q)
        orders = {
            "queue order": np.arange(n),
            "random": rng.permutation(n),
            "octant, origin": np.argsort((octant << 30) | m, kind="stable"),
            "origin, octant": np.argsort((m << 3) | octant, kind="stable"),
        }
        ctx.trace(rays[:65536])
        ctx.trace(rays[:65536])
        ref = None
        for name, perm in orders.items():
            h = ctx.trace(np.ascontiguousarray(rays[perm]))
            back = np.empty_like(h)
            back[perm] = h
            if ref is None:
                ref = back
            assert np.array_equal(back, ref), name  # the hits do not depend on the order
            print("bounce %d: %-16s %9d rays" % (bounce, name, n), flush=True)
