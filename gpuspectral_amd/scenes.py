"""Synthetic scene builders (numpy only) for the BASELINE.json configs.

"bathroom2" and the test scenes' sphere meshes are not in the reference tree
and there is no network, so the benchmark scenes are generated procedurally and
deterministically (seeded).  Scenes are assembled the way the reference loader
assembles them (S/engine/Loader.cpp:253-349): render objects referencing shared
de-indexed meshes, a material per object, one TriangleLight per triangle of an
emitting object, transformed on the host with glm's mat4*vec4 association.
"""
import functools
import math

import numpy as np

from . import abi

F = np.float32


def _glm_mul_point(m16, p):
    m = np.asarray(m16, F).reshape(4, 4)  # m[c] = column c
    x, y, z = F(p[0]), F(p[1]), F(p[2])
    return ((m[0] * x + m[1] * y) + (m[2] * z + m[3] * F(1.0))).astype(F)


def trs(translate=(0, 0, 0), scale=(1, 1, 1), rot_y_deg=0.0):
    """Column-major (glm memory order) T * Ry * S as 16 float32."""
    c, s = math.cos(math.radians(rot_y_deg)), math.sin(math.radians(rot_y_deg))
    sx, sy, sz = (scale, scale, scale) if np.isscalar(scale) else scale
    m = np.array(
        [
            [c * sx, 0.0, s * sz, translate[0]],
            [0.0, sy, 0.0, translate[1]],
            [-s * sx, 0.0, c * sz, translate[2]],
            [0.0, 0.0, 0.0, 1.0],
        ],
        np.float64,
    )
    return m.T.astype(F).reshape(16).copy()


def rowmajor(values16):
    """Mitsuba <matrix value="..."> (row-major) -> glm memory order."""
    return np.asarray(values16, np.float64).reshape(4, 4).T.astype(F).reshape(16).copy()


# ---- meshes (de-indexed: 3 consecutive vertices = 1 triangle) --------------------
def rect_mesh():
    """S/assets/rect.obj: unit rectangle in z = 0, normal +z, faces 1 3 2 / 3 4 2."""
    v = np.array([(-1, 1, 0), (1, 1, 0), (-1, -1, 0), (1, -1, 0)], F)
    idx = [0, 2, 1, 2, 3, 1]
    return v[idx].copy(), np.tile(np.array([0, 0, 1], F), (6, 1))


def box_mesh():
    """S/assets/box.obj: cube of half-extent 1, outward normals, rect.obj winding per face."""
    faces = [
        ((1, 0, 0), [(1, 1, 1), (1, 1, -1), (1, -1, 1), (1, -1, -1)]),
        ((-1, 0, 0), [(-1, 1, -1), (-1, 1, 1), (-1, -1, -1), (-1, -1, 1)]),
        ((0, 1, 0), [(-1, 1, -1), (1, 1, -1), (-1, 1, 1), (1, 1, 1)]),
        ((0, -1, 0), [(-1, -1, 1), (1, -1, 1), (-1, -1, -1), (1, -1, -1)]),
        ((0, 0, 1), [(-1, 1, 1), (1, 1, 1), (-1, -1, 1), (1, -1, 1)]),
        ((0, 0, -1), [(1, 1, -1), (-1, 1, -1), (1, -1, -1), (-1, -1, -1)]),
    ]
    ps, ns = [], []
    for n, c in faces:
        c = np.array(c, F)
        ps.append(c[[0, 2, 1, 2, 3, 1]])
        ns.append(np.tile(np.array(n, F), (6, 1)))
    return np.concatenate(ps), np.concatenate(ns)


def grid_mesh(fn, nu, nv, wrap_u=False):
    """Tessellate a parametric surface fn(u, v) -> (pos[...,3], nrm[...,3]), u,v in [0,1]."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    pos, nrm = fn(uu, vv)
    pos = pos.astype(F)
    nrm = nrm.astype(F)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    i, j = i.ravel(), j.ravel()
    a = (i, j)
    b = (i + 1, j)
    c = (i, j + 1)
    d = (i + 1, j + 1)
    tri = [a, b, c, b, d, c]
    P = np.stack([pos[t] for t in tri], axis=1).reshape(-1, 3)
    N = np.stack([nrm[t] for t in tri], axis=1).reshape(-1, 3)
    # drop zero-area triangles (poles)
    T = P.reshape(-1, 3, 3)
    area = np.linalg.norm(np.cross(T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]), axis=1)
    keep = np.repeat(area > 1e-12, 3)
    return np.ascontiguousarray(P[keep], F), np.ascontiguousarray(N[keep], F)


def sphere_mesh(nu=64, nv=32, bump=0.0, bump_freq=6.0, seed=0):
    """Unit UV sphere (outward winding, smooth normals); optional radial bumps."""
    rng = np.random.RandomState(seed)
    ph = rng.uniform(0, 2 * math.pi, 3)

    def fn(u, v):
        phi = 2 * math.pi * u
        th = math.pi * v
        d = np.stack([np.sin(th) * np.cos(phi), np.cos(th), -np.sin(th) * np.sin(phi)], -1)
        r = 1.0 + bump * (
            np.sin(bump_freq * d[..., 0] + ph[0]) * np.sin(bump_freq * d[..., 1] + ph[1]) * np.sin(bump_freq * d[..., 2] + ph[2])
        )
        return d * r[..., None], d

    return grid_mesh(fn, nu, nv)


def torus_mesh(nu=96, nv=48, r_minor=0.35):
    def fn(u, v):
        a = 2 * math.pi * u
        b = 2 * math.pi * v
        cx, cz = np.cos(a), -np.sin(a)
        n = np.stack([cx * np.cos(b), np.sin(b), cz * np.cos(b)], -1)
        p = np.stack([cx, np.zeros_like(cx), cz], -1) + r_minor * n
        return p, n

    return grid_mesh(fn, nu, nv)


def heightfield_mesh(n=128, amp=0.05, freq=5.0, seed=0):
    """[-1,1]^2 sheet in the xz plane, normal +y, with smooth waves."""
    rng = np.random.RandomState(seed)
    ph = rng.uniform(0, 2 * math.pi, 4)

    def fn(u, v):
        x = 2 * u - 1
        z = 1 - 2 * v
        y = amp * (np.sin(freq * x + ph[0]) * np.cos(freq * z + ph[1]) + 0.5 * np.sin(2.3 * freq * x + ph[2]) * np.sin(1.7 * freq * z + ph[3]))
        dydx = amp * freq * (np.cos(freq * x + ph[0]) * np.cos(freq * z + ph[1]) + 0.5 * 2.3 * np.cos(2.3 * freq * x + ph[2]) * np.sin(1.7 * freq * z + ph[3]))
        dydz = amp * freq * (-np.sin(freq * x + ph[0]) * np.sin(freq * z + ph[1]) + 0.5 * 1.7 * np.sin(2.3 * freq * x + ph[2]) * np.cos(1.7 * freq * z + ph[3]))
        nrm = np.stack([-dydx, np.ones_like(x), -dydz], -1)
        nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
        return np.stack([x, y, z], -1), nrm

    return grid_mesh(fn, n, n)


class SceneBuilder:
    """Mirrors the reference's Scene API (addMaterial / add<X>BSDF / addRenderObject)."""

    def __init__(self):
        self._pos, self._nrm = [], []
        self._nverts = 0
        self._instances = []
        self._bsdfs = [[] for _ in range(abi.BSDF_TYPE_COUNT)]
        self._lights = []
        self.to_world = np.eye(4, dtype=F).reshape(16)
        self.fov = F(0.6)

    def add_mesh(self, pos, nrm):
        pos = np.ascontiguousarray(pos, F).reshape(-1, 3)
        nrm = np.ascontiguousarray(nrm, F).reshape(-1, 3)
        assert len(pos) == len(nrm) and len(pos) % 3 == 0
        rec = (self._nverts, len(pos))
        self._pos.append(pos)
        self._nrm.append(nrm)
        self._nverts += len(pos)
        return rec

    def add_bsdf(self, btype, **fields):
        rec = np.zeros(1, abi.BSDF_DTYPES[btype])
        for k, v in fields.items():
            rec[k] = v
        self._bsdfs[btype].append(rec)
        return abi.bsdf_handle(btype, len(self._bsdfs[btype]) - 1)

    # the material conversions of S/engine/Loader.cpp:145-234
    def diffuse(self, rgb):
        return self.add_bsdf(abi.BSDF_DIFFUSE, reflectance=rgb)

    def dielectric(self, int_ior=1.5, ext_ior=1.0):
        return self.add_bsdf(abi.BSDF_SMOOTH_DIELECTRIC, ior_in=int_ior, ior_out=ext_ior)

    def mirror(self, eta=0.0):
        return self.add_bsdf(abi.BSDF_SMOOTH_CONDUCTOR, ior_in=eta, ior_out=1.0)

    def plastic(self, rgb, int_ior=1.3):
        ior = F(int_ior)
        r0 = (ior - F(1)) / (ior + F(1))
        return self.add_bsdf(abi.BSDF_SMOOTH_PLASTIC, diffuse=rgb, ior_in=ior, ior_out=1.0, r0=F(r0 * r0))

    def rough_plastic(self, rgb, alpha=0.05, int_ior=1.3):
        ior = F(int_ior)
        r0 = (ior - F(1)) / (ior + F(1))
        return self.add_bsdf(
            abi.BSDF_ROUGH_PLASTIC, diffuse=rgb, ior_in=ior, ior_out=1.0, r0=F(r0 * r0), alpha=F(F(math.sqrt(2.0)) * F(alpha))
        )

    def rough_conductor(self, eta, k, alpha=0.1, reflectance=(1, 1, 1)):
        return self.add_bsdf(
            abi.BSDF_ROUGH_CONDUCTOR, eta=eta, k=k, reflectance=reflectance, alpha=F(F(math.sqrt(2.0)) * F(alpha))
        )

    def smooth_floor(self, rgb, r0=0.04):
        return self.add_bsdf(abi.BSDF_SMOOTH_FLOOR, diffuse=rgb, r0=r0)

    def rough_floor(self, rgb, r0=0.04, alpha=0.1):
        return self.add_bsdf(abi.BSDF_ROUGH_FLOOR, diffuse=rgb, r0=r0, alpha=alpha)

    def add_object(self, mesh, transform, bsdf, twofaced=True, emission=None):
        first, count = mesh
        inst = np.zeros(1, abi.INSTANCE_DT)
        inst["transform"] = np.asarray(transform, F).reshape(16)
        inst["bsdf"] = bsdf
        inst["twofaced"] = 1 if twofaced else 0
        inst["first_vertex"] = first
        inst["vertex_count"] = count
        if emission is not None:
            inst["emission"] = np.asarray(emission, F)
            pos = np.concatenate(self._pos)[first : first + count]
            lt = np.zeros(count // 3, abi.LIGHT_DT)
            m = inst["transform"][0]
            for i in range(count // 3):
                for k in range(3):
                    lt["positions"][i, k] = _glm_mul_point(m, pos[3 * i + k])
            lt["radiance"][:, :3] = np.asarray(emission, F)
            lt["radiance"][:, 3] = 1.0
            self._lights.append(lt)
        self._instances.append(inst)

    def camera_lookat(self, eye, target, up=(0, 1, 0), fov_deg=40.0):
        """Sensor matrix for the reference's ray recipe (raygen.rgen:20-35): the shader forms
        d = M * normalize(-x, y, z) and then negates d.y IN WORLD SPACE, so the matrix that
        yields right*x + down*y + forward*z is [left | down | forward] with its y row negated
        (the Cornell XML matrix is exactly that for an unpitched camera)."""
        eye, target, up = (np.asarray(a, np.float64) for a in (eye, target, up))
        fwd = target - eye
        fwd /= np.linalg.norm(fwd)
        left = np.cross(up, fwd)
        left /= np.linalg.norm(left)
        newup = np.cross(fwd, left)
        m = np.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2] = left, -newup, fwd
        m[1, :3] *= -1.0
        m[:3, 3] = eye
        self.to_world = m.T.astype(F).reshape(16).copy()
        self.fov = F(np.float64(F(fov_deg)) * math.pi / 180.0)

    def build(self):
        sc = abi.SceneArrays()
        sc.instances = np.concatenate(self._instances) if self._instances else np.zeros(0, abi.INSTANCE_DT)
        sc.positions = np.concatenate(self._pos) if self._pos else np.zeros((0, 3), F)
        sc.normals = np.concatenate(self._nrm) if self._nrm else np.zeros((0, 3), F)
        sc.bsdfs = [np.concatenate(b) if b else np.zeros(0, dt) for b, dt in zip(self._bsdfs, abi.BSDF_DTYPES)]
        sc.lights = np.concatenate(self._lights) if self._lights else np.zeros(0, abi.LIGHT_DT)
        sc.to_world = np.asarray(self.to_world, F).reshape(16).copy()
        sc.fov = F(self.fov)
        return sc


# Cornell transforms of S/assets/scenes/cornell-box/scene.xml:66-116 (row-major text values)
_CORNELL = {
    "floor": [-4.37114e-08, 1, 4.37114e-08, 0, 0, -8.74228e-08, 2, 0, 1, 4.37114e-08, 1.91069e-15, 0, 0, 0, 0, 1],
    "ceiling": [-1, 7.64274e-15, -1.74846e-07, 0, 8.74228e-08, 8.74228e-08, -2, 2, 0, -1, -4.37114e-08, 0, 0, 0, 0, 1],
    "back": [1.91069e-15, 1, 1.31134e-07, 0, 1, 3.82137e-15, -8.74228e-08, 1, -4.37114e-08, 1.31134e-07, -2, -1, 0, 0, 0, 1],
    "right": [4.37114e-08, -1.74846e-07, 2, 1, 1, 3.82137e-15, -8.74228e-08, 1, 3.82137e-15, 1, 2.18557e-07, 0, 0, 0, 0, 1],
    "left": [-4.37114e-08, 8.74228e-08, -2, -1, 1, 3.82137e-15, -8.74228e-08, 1, 0, -1, -4.37114e-08, 0, 0, 0, 0, 1],
    "light": [0.235, -1.66103e-08, -7.80685e-09, -0.005, -2.05444e-08, 3.90343e-09, -0.0893, 1.98, 2.05444e-08, 0.19, 8.30516e-09, -0.03, 0, 0, 0, 1],
    "camera": [-1, 0, 0, 0, 0, 1, 0, 1, 0, 0, -1, 6.8, 0, 0, 0, 1],
}

GOLD_ETA = (1.65746, 0.880369, 0.521229)  # S/assets/scenes/test3/scene.xml:93-94
GOLD_K = (9.22387, 6.26952, 4.837)


def _cornell_room(b, light_radiance=(17, 12, 4), floor_bsdf=None):
    rect = b.add_mesh(*rect_mesh())
    white = b.diffuse((0.725, 0.71, 0.68))
    b.add_object(rect, rowmajor(_CORNELL["floor"]), floor_bsdf if floor_bsdf is not None else white)
    b.add_object(rect, rowmajor(_CORNELL["ceiling"]), white)
    b.add_object(rect, rowmajor(_CORNELL["back"]), white)
    b.add_object(rect, rowmajor(_CORNELL["right"]), b.diffuse((0.14, 0.45, 0.091)))
    b.add_object(rect, rowmajor(_CORNELL["left"]), b.diffuse((0.63, 0.065, 0.05)))
    b.to_world = rowmajor(_CORNELL["camera"])
    b.fov = F(np.float64(F(19.5)) * math.pi / 180.0)
    return rect, white


def cornell_materials(sphere_res=48, full=True):
    """BASELINE config 2: Cornell room + generated spheres/boxes carrying the full BSDF set
    (modelled on S/assets/scenes/test3/scene.xml:56-105,165-178)."""
    b = SceneBuilder()
    floor = b.rough_floor((0.6, 0.6, 0.6), 0.04, 0.2) if full else None
    rect, white = _cornell_room(b, floor_bsdf=floor)
    sph = b.add_mesh(*sphere_mesh(2 * sphere_res, sphere_res))
    box = b.add_mesh(*box_mesh())
    b.add_object(sph, trs((0.55, 0.35, -0.1), 0.35), b.rough_conductor(GOLD_ETA, GOLD_K, 0.1))
    b.add_object(sph, trs((-0.55, 0.35, 0.05), 0.35), b.dielectric(1.3, 1.0), twofaced=False)
    b.add_object(sph, trs((0.0, 0.22, 0.5), 0.22), b.rough_plastic((1, 0.578676, 0.134734), 0.05, 1.3))
    if full:
        b.add_object(box, trs((-0.05, 0.5, -0.55), (0.2, 0.5, 0.2), 20.0), b.plastic((0.434734, 0.578676, 1.0), 1.3))
        b.add_object(box, trs((0.62, 1.2, -0.7), (0.25, 0.25, 0.02), -15.0), b.mirror(0.0))
        b.add_object(sph, trs((-0.6, 1.3, -0.6), 0.18), b.smooth_floor((0.2, 0.5, 0.7), 0.05))
    b.add_object(rect, rowmajor(_CORNELL["light"]), b.diffuse((0, 0, 0)), twofaced=True, emission=(17, 12, 4))
    return b.build()


def interior(target_tris=600_000, seed=7):
    """BASELINE configs 3/4: seeded procedural stand-in for Mitsuba 'bathroom2'
    (closed room, tessellated fixtures; rough conductor + glass + diffuse; emissive panels)."""
    rng = np.random.RandomState(seed)
    b = SceneBuilder()
    rect = b.add_mesh(*rect_mesh())
    W, H, D = 4.0, 2.6, 5.0  # half extents x, full height, half depth
    wall = b.diffuse((0.78, 0.76, 0.72))
    tile = b.diffuse((0.55, 0.62, 0.66))
    # room shell: floor is a wavy heightfield (most of the floor-level triangles)
    budget = max(target_tris - 12 * 2 - 8, 1000)
    n_floor = int(math.sqrt(0.10 * budget / 2))
    floor = b.add_mesh(*heightfield_mesh(n_floor, amp=0.004, freq=9.0, seed=seed))
    b.add_object(floor, trs((0, 0, 0), (W, 1.0, D)), tile)
    b.add_object(rect, rowmajor([W, 0, 0, 0, 0, 0, -1, H, 0, D, 0, 0, 0, 0, 0, 1]), wall)  # ceiling (normal -y)
    b.add_object(rect, rowmajor([W, 0, 0, 0, 0, H / 2, 0, H / 2, 0, 0, 1, -D, 0, 0, 0, 1]), wall)  # back z=-D
    b.add_object(rect, rowmajor([-W, 0, 0, 0, 0, H / 2, 0, H / 2, 0, 0, -1, D, 0, 0, 0, 1]), wall)  # front z=+D
    b.add_object(rect, rowmajor([0, 0, 1, -W, 0, H / 2, 0, H / 2, -D, 0, 0, 0, 0, 0, 0, 1]), b.diffuse((0.7, 0.5, 0.4)))
    b.add_object(rect, rowmajor([0, 0, -1, W, 0, H / 2, 0, H / 2, D, 0, 0, 0, 0, 0, 0, 1]), b.diffuse((0.4, 0.55, 0.7)))
    # fixtures
    remaining = budget - 2 * n_floor * n_floor
    n_obj = 36
    per = remaining // n_obj
    chrome = b.rough_conductor((2.8, 2.9, 2.7), (3.3, 3.2, 2.9), 0.08)
    brass = b.rough_conductor(GOLD_ETA, GOLD_K, 0.15)
    glass = b.dielectric(1.5, 1.0)
    ceramic = b.rough_plastic((0.85, 0.85, 0.82), 0.08, 1.5)
    mats = [chrome, brass, glass, ceramic, b.diffuse((0.6, 0.3, 0.25)), b.diffuse((0.3, 0.5, 0.35))]
    meshes = []
    nv = max(8, int(math.sqrt(per / 4)))
    for i in range(6):
        meshes.append(b.add_mesh(*sphere_mesh(2 * nv, nv, bump=0.06 * (i % 3), bump_freq=5 + 2 * i, seed=seed + i)))
    nt = max(8, int(math.sqrt(per / 4)))
    meshes.append(b.add_mesh(*torus_mesh(2 * nt, nt, 0.3)))
    gx, gz = 6, 6
    k = 0
    for ix in range(gx):
        for iz in range(gz):
            cx = -W + (ix + 0.5) * (2 * W / gx) + rng.uniform(-0.15, 0.15)
            cz = -D + (iz + 0.5) * (2 * D / gz) + rng.uniform(-0.2, 0.2)
            r = rng.uniform(0.18, 0.32)
            y = r + rng.uniform(0.0, 1.2) * (k % 3 == 0)
            m = meshes[k % len(meshes)]
            mat = mats[k % len(mats)]
            b.add_object(m, trs((cx, y + 0.02, cz), r, rng.uniform(0, 360)), mat, twofaced=(mat != glass))
            k += 1
    # emissive ceiling panels
    for (lx, lz) in [(-2.0, -2.5), (2.0, -2.5), (-2.0, 2.0), (2.0, 2.0)]:
        b.add_object(
            rect,
            rowmajor([0.45, 0, 0, lx, 0, 0, -1, H - 0.01, 0, 0.45, 0, lz, 0, 0, 0, 1]),
            b.diffuse((0, 0, 0)),
            emission=(14.0, 13.0, 11.0),
        )
    b.camera_lookat((3.2, 1.5, 4.4), (-0.6, 0.7, -1.0), fov_deg=55.0)
    return b.build()


def caustics(target_tris=200_000, seed=11):
    """BASELINE config 5: dielectric-heavy scene (nested / adjacent glass solids over a diffuse
    floor, small bright emitter): deep all-delta paths, no NEE on most vertices."""
    rng = np.random.RandomState(seed)
    b = SceneBuilder()
    rect = b.add_mesh(*rect_mesh())
    floor = b.diffuse((0.7, 0.7, 0.7))
    b.add_object(rect, rowmajor([6, 0, 0, 0, 0, 0, 1, 0, 0, -6, 0, 0, 0, 0, 0, 1]), floor)
    b.add_object(rect, rowmajor([6, 0, 0, 0, 0, 3, 0, 3, 0, 0, 1, -6, 0, 0, 0, 1]), b.diffuse((0.5, 0.5, 0.6)))
    n_obj = 25
    per = max(target_tris // n_obj, 200)
    nv = max(8, int(math.sqrt(per / 4)))
    meshes = [b.add_mesh(*sphere_mesh(2 * nv, nv, bump=0.08 * i, bump_freq=4 + i, seed=seed + i)) for i in range(4)]
    glasses = [b.dielectric(1.3 + 0.05 * i, 1.0) for i in range(5)]
    k = 0
    for ix in range(5):
        for iz in range(5):
            r = rng.uniform(0.3, 0.5)
            b.add_object(
                meshes[k % 4],
                trs((-2.4 + 1.2 * ix + rng.uniform(-0.1, 0.1), r + 0.01, -2.4 + 1.2 * iz + rng.uniform(-0.1, 0.1)), r, rng.uniform(0, 360)),
                glasses[k % 5],
                twofaced=False,
            )
            k += 1
    b.add_object(
        rect,
        rowmajor([0.3, 0, 0, 0.5, 0, 0, -1, 4.0, 0, 0.3, 0, 0.5, 0, 0, 0, 1]),
        b.diffuse((0, 0, 0)),
        emission=(400.0, 380.0, 340.0),
    )
    b.camera_lookat((0.0, 3.2, 6.5), (0.0, 0.4, 0.0), fov_deg=45.0)
    return b.build()


@functools.lru_cache(maxsize=64)
def _tile_pixel_ids_cached(width, height, rank, world, tile):
    ty, tx = np.meshgrid(np.arange((height + tile - 1) // tile), np.arange((width + tile - 1) // tile), indexing="ij")
    mine = (ty * tx.shape[1] + tx + ty) % world == rank  # +ty staggers columns so ranks interleave in both axes
    mask = np.repeat(np.repeat(mine, tile, axis=0), tile, axis=1)[:height, :width]
    ids = np.flatnonzero(mask).astype(np.uint32)  # row-major scan = increasing global pixel index
    ids.setflags(write=False)
    return ids


def tile_pixel_ids(width, height, rank, world, tile=32):
    """Pixels of the tiles owned by `rank` when tile x tile blocks of the frame are dealt
    round-robin to `world` ranks; returned sorted (strictly increasing).  The result is
    cached and read-only (bench.py's timed gather asks for every rank's list)."""
    return _tile_pixel_ids_cached(int(width), int(height), int(rank), int(world), int(tile))
