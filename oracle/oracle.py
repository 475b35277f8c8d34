"""oracle/oracle.py -- ctypes binding of liboracle_pt.so. TEST INFRASTRUCTURE ONLY.

See oracle/README.md for what the oracle restates and how far it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from gpuspectral_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def usable_cpus():
    """Host threads this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU boxes
    show 256 hardware threads but grant 16 CPUs of time; oversubscribing a quota only adds throttling stalls)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


class OracleStats(C.Structure):
    _fields_ = [
        ("extension_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("shaded_vertices", C.c_uint64),
        ("samples", C.c_uint64),
        ("nodes_visited", C.c_uint64),
        ("tris_tested", C.c_uint64),
        ("stat_rays", C.c_uint64),
        ("seconds", C.c_double),
        ("num_triangles", C.c_uint64),
        ("num_bvh_nodes", C.c_uint64),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def build(force=False):
    """Compile liboracle_pt.so with the committed Makefile (gcc only)."""
    so = os.path.join(_HERE, "liboracle_pt.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "liboracle_pt.so"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.environ.get("ORACLE_LIB_PATH") or os.path.join(_HERE, "liboracle_pt.so")  # (override: the ASan/UBSan build)
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.POINTER(abi.SceneDesc)]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_build_seconds.restype = C.c_double
        L.oracle_build_seconds.argtypes = [C.c_void_p]
        L.oracle_render.restype = C.c_int
        L.oracle_render.argtypes = [
            C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64,
            C.POINTER(abi.RenderParams), C.c_void_p, C.c_int, C.c_int, C.POINTER(OracleStats),
        ]
        L.oracle_trace.restype = C.c_int
        L.oracle_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
        L.oracle_ray_log.argtypes = [C.c_int]
        L.oracle_ray_log.restype = None
        L.oracle_ray_log_read.argtypes = [C.c_void_p, C.c_uint64]
        L.oracle_ray_log_read.restype = C.c_uint64
        L.oracle_primary_ray.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_tea.restype = C.c_uint32
        L.oracle_tea.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_pcg_hash.restype = C.c_uint32
        L.oracle_pcg_hash.argtypes = [C.c_uint32]
        L.oracle_rand_pcg.restype = C.c_uint32
        L.oracle_rand_pcg.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p]
        L.oracle_rand_uniform.restype = C.c_float
        L.oracle_rand_uniform.argtypes = [C.c_uint32]
        L.oracle_bsdf_sample.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_bsdf_eval.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_sample_light.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_det_math.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_transform_inv_t.argtypes = [C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


HIT_DT = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim", "<i4")])


class Oracle:
    """Scalar CPU integrator over one flattened scene (abi.SceneArrays)."""

    def __init__(self, scene):
        self.scene = scene
        self._desc = scene.desc()
        self._h = lib().oracle_create(C.byref(self._desc))
        if not self._h:
            raise RuntimeError("oracle_create failed")

    def close(self):
        if self._h:
            lib().oracle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def build_seconds(self):
        return lib().oracle_build_seconds(self._h)

    def render(self, width, height, spp=1, first_timestamp=0, accum=None, pixel_ids=None, threads=0,
               params=None, collect_traversal_stats=False):
        """Returns (accum[n,4] float32, stats dict).  accum is updated in place when given."""
        p = params or abi.default_render_params(spp, first_timestamp)
        p.spp, p.first_timestamp = spp, first_timestamp
        if threads <= 0:
            threads = usable_cpus()
        if pixel_ids is not None:
            pixel_ids = np.ascontiguousarray(pixel_ids, np.uint32)
            n = len(pixel_ids)
        else:
            n = width * height
        if accum is None:
            accum = np.zeros((n, 4), np.float32)
        assert accum.dtype == np.float32 and accum.size == n * 4 and accum.flags.c_contiguous
        st = OracleStats()
        rc = lib().oracle_render(
            self._h, width, height, pixel_ids.ctypes.data if pixel_ids is not None else None, n,
            C.byref(p), accum.ctypes.data, threads, 1 if collect_traversal_stats else 0, C.byref(st),
        )
        if rc != 0:
            raise RuntimeError("oracle_render failed: %d" % rc)
        return accum, st.as_dict()

    def extension_rays_of(self, width, height, spp=1, first_timestamp=0, pixel_ids=None, params=None):
        """Debug aid: the extension rays {o, tmin, d, tmax} the oracle traces for this render, in trace order (one thread)."""
        lib().oracle_ray_log(1)
        try:
            self.render(width, height, spp=spp, first_timestamp=first_timestamp, pixel_ids=pixel_ids, threads=1, params=params)
            n = lib().oracle_ray_log_read(None, 0)
            rays = np.zeros((n, 8), np.float32)
            lib().oracle_ray_log_read(rays.ctypes.data, n)
        finally:
            lib().oracle_ray_log(0)
        return rays

    def trace(self, rays, any_hit=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros(len(rays), HIT_DT)
        lib().oracle_trace(self._h, rays.ctypes.data, len(rays), 1 if any_hit else 0, hits.ctypes.data)
        return hits

    def primary_ray(self, width, height, px, py):
        out = np.zeros(6, np.float32)
        lib().oracle_primary_ray(self._h, width, height, px, py, out.ctypes.data)
        return out

    def bsdf_sample(self, handle, wo, seed):
        wo = np.ascontiguousarray(wo, np.float32)
        out = np.zeros(9, np.float32)
        lib().oracle_bsdf_sample(self._h, handle, wo.ctypes.data, seed, out.ctypes.data)
        return out

    def bsdf_eval(self, handle, wo, wi):
        wo = np.ascontiguousarray(wo, np.float32)
        wi = np.ascontiguousarray(wi, np.float32)
        out = np.zeros(5, np.float32)
        lib().oracle_bsdf_eval(self._h, handle, wo.ctypes.data, wi.ctypes.data, out.ctypes.data)
        return out

    def sample_light(self, pos, seed):
        pos = np.ascontiguousarray(pos, np.float32)
        out = np.zeros(8, np.float32)
        lib().oracle_sample_light(self._h, pos.ctypes.data, seed, out.ctypes.data)
        return out


def tea(a, b):
    return lib().oracle_tea(a, b)


def pcg_hash(v):
    return lib().oracle_pcg_hash(v)


def rand_pcg(seed, n):
    out = np.zeros(n, np.uint32)
    final = lib().oracle_rand_pcg(seed, n, out.ctypes.data)
    return out, final


def rand_uniform(seed):
    return lib().oracle_rand_uniform(seed)


def det_math(x):
    x = np.ascontiguousarray(x, np.float32)
    s, c, lg, ex = (np.zeros_like(x) for _ in range(4))
    lib().oracle_det_math(x.ctypes.data, x.size, s.ctypes.data, c.ctypes.data, lg.ctypes.data, ex.ctypes.data)
    return s, c, lg, ex


def transform_inv_t(m16):
    m = np.ascontiguousarray(m16, np.float32).reshape(16)
    out = np.zeros(16, np.float32)
    lib().oracle_transform_inv_t(m.ctypes.data, out.ctypes.data)
    return out
