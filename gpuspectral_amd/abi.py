"""ctypes / numpy mirror of ``include/gpuspectral_pt.h`` (the C-ABI boundary).

Nothing here computes anything: it only describes the POD layouts so that
Python callers (tests, bench.py) can hand arrays to the C ABI.  Layouts follow
the reference's scalar structs (S/renderer/Scene.h:29-109).
"""
import ctypes as C

import numpy as np

GSP_ABI_VERSION = 8

BSDF_DIFFUSE = 0
BSDF_SMOOTH_DIELECTRIC = 1
BSDF_SMOOTH_CONDUCTOR = 2
BSDF_SMOOTH_PLASTIC = 3
BSDF_ROUGH_CONDUCTOR = 4
BSDF_SMOOTH_FLOOR = 5
BSDF_ROUGH_FLOOR = 6
BSDF_ROUGH_PLASTIC = 7
BSDF_TYPE_COUNT = 8

BSDF_NAMES = (
    "diffuse",
    "smooth_dielectric",
    "smooth_conductor",
    "smooth_plastic",
    "rough_conductor",
    "smooth_floor",
    "rough_floor",
    "rough_plastic",
)


def bsdf_handle(btype, index):
    """(type << 16) | index -- S/renderer/Scene.h:83-97."""
    return ((int(btype) & 0xFFFF) << 16) | (int(index) & 0xFFFF)


# numpy record layouts (all tightly packed, 4-byte scalars)
DIFFUSE_DT = np.dtype([("reflectance", "<f4", 3), ("has_texture", "<i4")])
SMOOTH_DIELECTRIC_DT = np.dtype([("ior_in", "<f4"), ("ior_out", "<f4")])
SMOOTH_CONDUCTOR_DT = np.dtype([("ior_in", "<f4"), ("ior_out", "<f4")])
SMOOTH_PLASTIC_DT = np.dtype([("diffuse", "<f4", 3), ("ior_in", "<f4"), ("ior_out", "<f4"), ("r0", "<f4")])
ROUGH_CONDUCTOR_DT = np.dtype(
    [("eta", "<f4", 3), ("k", "<f4", 3), ("reflectance", "<f4", 3), ("alpha", "<f4"), ("has_texture", "<i4")]
)
SMOOTH_FLOOR_DT = np.dtype([("diffuse", "<f4", 3), ("r0", "<f4")])
ROUGH_FLOOR_DT = np.dtype([("diffuse", "<f4", 3), ("r0", "<f4"), ("alpha", "<f4")])
ROUGH_PLASTIC_DT = np.dtype(
    [
        ("diffuse", "<f4", 3),
        ("ior_in", "<f4"),
        ("ior_out", "<f4"),
        ("r0", "<f4"),
        ("alpha", "<f4"),
        ("has_texture", "<i4"),
    ]
)
BSDF_DTYPES = (
    DIFFUSE_DT,
    SMOOTH_DIELECTRIC_DT,
    SMOOTH_CONDUCTOR_DT,
    SMOOTH_PLASTIC_DT,
    ROUGH_CONDUCTOR_DT,
    SMOOTH_FLOOR_DT,
    ROUGH_FLOOR_DT,
    ROUGH_PLASTIC_DT,
)
BSDF_SIZES = (16, 8, 8, 24, 44, 16, 20, 32)
assert tuple(d.itemsize for d in BSDF_DTYPES) == BSDF_SIZES

LIGHT_DT = np.dtype([("positions", "<f4", (3, 4)), ("radiance", "<f4", 4)])
assert LIGHT_DT.itemsize == 64

INSTANCE_DT = np.dtype(
    [
        ("transform", "<f4", 16),
        ("emission", "<f4", 3),
        ("bsdf", "<u4"),
        ("twofaced", "<u4"),
        ("first_vertex", "<u4"),
        ("vertex_count", "<u4"),
    ]
)
assert INSTANCE_DT.itemsize == 92


class Camera(C.Structure):
    _fields_ = [("to_world", C.c_float * 16), ("fov", C.c_float)]


TEXTURE_DT = np.dtype([("width", "<u4"), ("height", "<u4"), ("first_texel", "<u8")])  # gsp_texture, 16 B


class Envmap(C.Structure):  # gsp_envmap
    _fields_ = [("texels", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("to_local", C.c_float * 16)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("instances", C.c_void_p),
        ("num_instances", C.c_uint32),
        ("positions", C.c_void_p),
        ("normals", C.c_void_p),
        ("num_vertices", C.c_uint64),
        ("diffuse_bsdfs", C.c_void_p),
        ("smooth_dielectric_bsdfs", C.c_void_p),
        ("smooth_conductor_bsdfs", C.c_void_p),
        ("smooth_plastic_bsdfs", C.c_void_p),
        ("rough_conductor_bsdfs", C.c_void_p),
        ("smooth_floor_bsdfs", C.c_void_p),
        ("rough_floor_bsdfs", C.c_void_p),
        ("rough_plastic_bsdfs", C.c_void_p),
        ("num_bsdfs", C.c_uint32 * BSDF_TYPE_COUNT),
        ("lights", C.c_void_p),
        ("num_lights", C.c_uint32),
        ("camera", Camera),
        # dormant-feature extension (textures / environment map); all zero = the reference's behaviour
        ("uvs", C.c_void_p),
        ("textures", C.c_void_p),
        ("num_textures", C.c_uint32),
        ("texels", C.c_void_p),
        ("num_texels", C.c_uint64),
        ("texel_decode", C.c_void_p),
        ("envmap", Envmap),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),  # (ABI 8) sizeof the host's struct; 0 = the ABI-8 layout
        ("spp", C.c_uint32),
        ("first_timestamp", C.c_uint32),
        ("max_depth", C.c_uint32),
        ("rr_start_depth", C.c_uint32),
        ("clamp", C.c_float),
        ("timestamps_in_flight", C.c_uint32),
        ("collect_traversal_stats", C.c_uint32),
        ("collect_kernel_times", C.c_uint32),
        ("disable_nee", C.c_uint32),  # (ABI 7) RenderParams.nee inverted, PathTracer.h:36-41; 0 = the shipped shader (`#define NEE true`)
    ]


def default_render_params(spp=1, first_timestamp=0):
    """The reference's shader literals: raygen.rgen:27,60,66; rayhit.rchit:656."""
    return RenderParams(C.sizeof(RenderParams), spp, first_timestamp, 50, 10, 20.0, 0, 0, 0, 0)


GATHER_AUTO, GATHER_RCCL, GATHER_COPY = 0, 1, 2


class CtxOptions(C.Structure):
    """gsp_ctx_options (ABI 5, refit_growth: 6); a field left 0 means "default"."""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("lanes", C.c_uint32),
        ("pool_paths", C.c_uint64),
        ("ring_bytes", C.c_uint64),
        ("memory_share", C.c_double),
        ("primary_memo", C.c_uint32),
        ("finish_paths", C.c_uint32),
        ("reinsert_rounds", C.c_uint32),
        ("gather_route", C.c_uint32),
        ("refit_growth", C.c_double),  # (ABI 6) 0 default (1.25); <= 1: gsp_update_instances always rebuilds
        ("geometry_versions", C.c_uint32),  # (r05) slots of the geometry ring: 0 default (as many as fit, <= 64), 1 = none
        ("reserved_", C.c_uint32),
    ]

    def __init__(self, **kw):
        super().__init__(**kw)
        self.struct_size = C.sizeof(CtxOptions)


def options_from_env(env=None):
    """TEST / A-B TOOLING: the GSP_* variables the probe scripts of earlier rounds set, mapped onto gsp_ctx_options (the
    library itself never reads the environment).  GSP_POOL_PATHS, GSP_RING_BYTES, GSP_MEMORY_SHARE, GSP_PRIMARY_MEMO=0|1,
    GSP_FINISH_PATHS (0 = never), GSP_LANES, GSP_BVH_REINSERT=rounds, GSP_MULTI_GATHER=rccl|copy."""
    import os

    e = os.environ if env is None else env
    o = CtxOptions()
    if "GSP_POOL_PATHS" in e:
        o.pool_paths = int(e["GSP_POOL_PATHS"])
    if "GSP_RING_BYTES" in e:
        o.ring_bytes = int(e["GSP_RING_BYTES"])
    if "GSP_MEMORY_SHARE" in e:
        o.memory_share = float(e["GSP_MEMORY_SHARE"])
    if "GSP_PRIMARY_MEMO" in e:
        o.primary_memo = 1 if int(e["GSP_PRIMARY_MEMO"]) else 2
    if "GSP_FINISH_PATHS" in e:
        o.finish_paths = int(e["GSP_FINISH_PATHS"]) or 0xFFFFFFFF
    if "GSP_LANES" in e:
        o.lanes = int(e["GSP_LANES"])
    if "GSP_BVH_REINSERT" in e:
        o.reinsert_rounds = int(e["GSP_BVH_REINSERT"]) + 1
    if "GSP_MULTI_GATHER" in e:
        o.gather_route = {"rccl": GATHER_RCCL, "copy": GATHER_COPY}.get(e["GSP_MULTI_GATHER"], GATHER_AUTO)
    return o


class Stats(C.Structure):
    _fields_ = [
        ("extension_rays", C.c_uint64),
        ("shadow_rays", C.c_uint64),
        ("shaded_vertices", C.c_uint64),
        ("samples", C.c_uint64),
        ("nodes_visited", C.c_uint64),
        ("tris_tested", C.c_uint64),
        ("stat_rays", C.c_uint64),
        ("shadow_nodes_visited", C.c_uint64),
        ("shadow_tris_tested", C.c_uint64),
        ("shadow_stat_rays", C.c_uint64),
        ("render_seconds", C.c_double),
        ("extend_kernel_ms", C.c_double),
        ("extend_launches", C.c_uint64),
        ("shade_kernel_ms", C.c_double),
        ("connect_kernel_ms", C.c_double),
        ("bvh_build_ms", C.c_double),
        ("num_triangles", C.c_uint64),
        ("num_bvh_nodes", C.c_uint64),
        ("device_bytes", C.c_uint64),
        ("algorithmic_bytes", C.c_uint64),
        ("memoised_rays", C.c_uint64),
        ("memo_build_rays", C.c_uint64),
        ("bvh_depth", C.c_uint64),
        ("nodes_from_lds", C.c_uint64),
        ("shadow_nodes_from_lds", C.c_uint64),
        ("shadow_stat_occluded", C.c_uint64),
        ("shadow_stat_occluded_nodes", C.c_uint64),
        ("scene_updates", C.c_uint64),
        ("scene_refits", C.c_uint64),  # (ABI 6)
        ("shadow_stat_no_triangle", C.c_uint64),  # (ABI 7)
        ("scene_drains", C.c_uint64),  # (ABI 7)
        ("scene_splits", C.c_uint64),  # (ABI 7)
    ]

    def as_dict(self):
        d = {name: getattr(self, name) for name, _ in self._fields_}
        # rays that were actually traced: path segments answered from the primary-hit memo are not
        d["traced_rays"] = d["extension_rays"] - d["memoised_rays"] + d["memo_build_rays"] + d["shadow_rays"]
        return d


class SceneArrays:
    """A flattened scene held in numpy arrays, convertible to ``SceneDesc``.

    This is the POD form of the reference's ``Scene`` (S/renderer/Scene.h:140-186)
    that crosses the C ABI.
    """

    def __init__(self):
        self.instances = np.zeros(0, INSTANCE_DT)
        self.positions = np.zeros((0, 3), np.float32)
        self.normals = np.zeros((0, 3), np.float32)
        self.bsdfs = [np.zeros(0, dt) for dt in BSDF_DTYPES]
        self.lights = np.zeros(0, LIGHT_DT)
        self.to_world = np.eye(4, dtype=np.float32).T.reshape(16).copy()  # glm memory order
        self.fov = np.float32(0.5)
        # dormant-feature extension (include/gpuspectral_pt.h): absent by default
        self.uvs = None            # (num_vertices, 2) float32
        self.textures = np.zeros(0, TEXTURE_DT)
        self.texels = np.zeros(0, np.uint32)     # RGBA8, all textures back to back
        self.texel_decode = None   # 256 float32, None = byte / 255
        self.env_texels = None     # (height, width, 4) float32, rows bottom-up
        self.env_to_local = np.eye(4, dtype=np.float32).reshape(16).copy()

    def add_texture(self, rgba8):
        """Append an (H, W, 4) uint8 image whose row 0 is the BOTTOM row; returns the has_texture value (index + 1)."""
        img = np.ascontiguousarray(rgba8, np.uint8)
        h, w = img.shape[:2]
        t = np.zeros(1, TEXTURE_DT)
        t["width"], t["height"], t["first_texel"] = w, h, len(self.texels)
        self.textures = np.concatenate([self.textures, t])
        self.texels = np.concatenate([self.texels, img.reshape(-1, 4).view("<u4").reshape(-1)])
        return len(self.textures)

    @property
    def num_triangles(self):
        return int(self.instances["vertex_count"].sum() // 3) if len(self.instances) else 0

    def desc(self):
        """Build a ``SceneDesc`` pointing at this object's arrays (keep `self` alive)."""
        self.instances = np.ascontiguousarray(self.instances, INSTANCE_DT)
        self.positions = np.ascontiguousarray(self.positions, np.float32).reshape(-1, 3)
        self.normals = np.ascontiguousarray(self.normals, np.float32).reshape(-1, 3)
        self.bsdfs = [np.ascontiguousarray(b, dt) for b, dt in zip(self.bsdfs, BSDF_DTYPES)]
        self.lights = np.ascontiguousarray(self.lights, LIGHT_DT)
        d = SceneDesc()
        d.instances = self.instances.ctypes.data
        d.num_instances = len(self.instances)
        d.positions = self.positions.ctypes.data
        d.normals = self.normals.ctypes.data
        d.num_vertices = len(self.positions)
        for name, arr in zip(BSDF_NAMES, self.bsdfs):
            setattr(d, name + "_bsdfs", arr.ctypes.data if len(arr) else None)
        for i, arr in enumerate(self.bsdfs):
            d.num_bsdfs[i] = len(arr)
        d.lights = self.lights.ctypes.data if len(self.lights) else None
        d.num_lights = len(self.lights)
        for i in range(16):
            d.camera.to_world[i] = float(self.to_world[i])
        d.camera.fov = float(self.fov)
        if len(self.textures):
            if self.uvs is None:
                self.uvs = np.zeros((len(self.positions), 2), np.float32)
            self.uvs = np.ascontiguousarray(self.uvs, np.float32).reshape(-1, 2)
            assert len(self.uvs) == len(self.positions)
            self.textures = np.ascontiguousarray(self.textures, TEXTURE_DT)
            self.texels = np.ascontiguousarray(self.texels, np.uint32)
            d.uvs = self.uvs.ctypes.data
            d.textures = self.textures.ctypes.data
            d.num_textures = len(self.textures)
            d.texels = self.texels.ctypes.data
            d.num_texels = len(self.texels)
            if self.texel_decode is not None:
                self.texel_decode = np.ascontiguousarray(self.texel_decode, np.float32)
                assert self.texel_decode.shape == (256,)
                d.texel_decode = self.texel_decode.ctypes.data
        if self.env_texels is not None:
            self.env_texels = np.ascontiguousarray(self.env_texels, np.float32)
            h, w = self.env_texels.shape[:2]
            d.envmap.texels = self.env_texels.ctypes.data
            d.envmap.width, d.envmap.height = w, h
            for i in range(16):
                d.envmap.to_local[i] = float(self.env_to_local[i])
        return d

    def save(self, path):
        """Write the flattened scene as one .npz (the POD arrays that cross the C ABI)."""
        np.savez_compressed(
            path, instances=self.instances, positions=self.positions, normals=self.normals, lights=self.lights,
            to_world=np.asarray(self.to_world, np.float32), fov=np.float32(self.fov),
            **{"bsdf_" + n: b for n, b in zip(BSDF_NAMES, self.bsdfs)}, **self._extension_arrays())

    def _extension_arrays(self):
        ext = {}
        if len(self.textures):
            ext.update(uvs=self.uvs, textures=self.textures, texels=self.texels)
            if self.texel_decode is not None:
                ext["texel_decode"] = self.texel_decode
        if self.env_texels is not None:
            ext.update(env_texels=self.env_texels, env_to_local=self.env_to_local)
        return ext

    @classmethod
    def load(cls, path):
        z = np.load(path)
        s = cls()
        s.instances = z["instances"].astype(INSTANCE_DT)
        s.positions, s.normals, s.lights = z["positions"], z["normals"], z["lights"].astype(LIGHT_DT)
        s.bsdfs = [z["bsdf_" + n].astype(dt) for n, dt in zip(BSDF_NAMES, BSDF_DTYPES)]
        s.to_world, s.fov = z["to_world"].copy(), np.float32(z["fov"])
        if "textures" in z:
            s.uvs, s.textures, s.texels = z["uvs"], z["textures"].astype(TEXTURE_DT), z["texels"]
            s.texel_decode = z["texel_decode"] if "texel_decode" in z else None
        if "env_texels" in z:
            s.env_texels, s.env_to_local = z["env_texels"], z["env_to_local"]
        return s
