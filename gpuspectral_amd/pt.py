"""ctypes binding of libgpuspectral_pt.so (the C ABI, include/gpuspectral_pt.h).

No computation happens here and there is no fallback: if the HIP library is
not built, or no GPU is present, the calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# every symbol include/gpuspectral_pt.h declares
EXPORTS = (
    "gsp_default_render_params",
    "gsp_abi_version",
    "gsp_build_info",
    "gsp_device_count",
    "gsp_ctx_create",
    "gsp_default_ctx_options",
    "gsp_ctx_create_ex",
    "gsp_ctx_destroy",
    "gsp_upload_scene",
    "gsp_update_camera",
    "gsp_update_instances",
    "gsp_update_tables",
    "gsp_frame_begin",
    "gsp_render",
    "gsp_sync",
    "gsp_download",
    "gsp_download_compact",
    "gsp_peek",
    "gsp_copy_accum_to_device",
    "gsp_peek_to_device",
    "gsp_upload_accum",
    "gsp_get_stats",
    "gsp_reset_stats",
    "gsp_trace",
    "gsp_debug_visit_histograms",
    "gsp_last_error",
    "gsp_tile_partition",
    "gsp_multi_create",
    "gsp_multi_create_ex",
    "gsp_multi_destroy",
    "gsp_multi_num_shares",
    "gsp_multi_upload_scene",
    "gsp_multi_update_camera",
    "gsp_multi_update_instances",
    "gsp_multi_update_tables",
    "gsp_multi_frame_begin",
    "gsp_multi_render",
    "gsp_multi_sync",
    "gsp_multi_gather",
    "gsp_multi_gather_route",
    "gsp_multi_download",
    "gsp_multi_get_stats",
    "gsp_multi_reset_stats",
    "gsp_multi_last_error",
)


class GspError(RuntimeError):
    pass


def lib_path():
    # GSP_LIB_PATH: developer override used to A/B kernel build variants (same ABI version only)
    return os.environ.get("GSP_LIB_PATH") or os.path.join(_HERE, "lib", "libgpuspectral_pt.so")


def load():
    """Load the HIP library; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise GspError(
            "HIP library %s not found: build it with `make -C gpuspectral_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback." % path
        )
    L = C.CDLL(path)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    L.gsp_default_render_params.argtypes = [C.POINTER(abi.RenderParams)]
    L.gsp_default_render_params.restype = None
    L.gsp_abi_version.restype = C.c_int
    L.gsp_build_info.restype = C.c_char_p
    L.gsp_device_count.restype = C.c_int
    L.gsp_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.gsp_default_ctx_options.argtypes = [C.POINTER(abi.CtxOptions)]
    L.gsp_default_ctx_options.restype = None
    L.gsp_ctx_create_ex.argtypes = [C.c_int, C.POINTER(abi.CtxOptions), C.POINTER(vp)]
    L.gsp_update_camera.argtypes = [vp, C.POINTER(abi.Camera)]
    L.gsp_update_instances.argtypes = [vp, vp, u32]
    L.gsp_update_tables.argtypes = [vp, C.POINTER(abi.SceneDesc)]
    L.gsp_ctx_destroy.argtypes = [vp]
    L.gsp_ctx_destroy.restype = None
    L.gsp_upload_scene.argtypes = [vp, C.POINTER(abi.SceneDesc)]
    L.gsp_frame_begin.argtypes = [vp, u32, u32, vp, u64]
    L.gsp_render.argtypes = [vp, C.POINTER(abi.RenderParams)]
    L.gsp_sync.argtypes = [vp]
    L.gsp_download.argtypes = [vp, vp]
    L.gsp_download_compact.argtypes = [vp, vp]
    L.gsp_peek.argtypes = [vp, vp, vp]
    L.gsp_copy_accum_to_device.argtypes = [vp, vp, u64]
    L.gsp_peek_to_device.argtypes = [vp, vp, u64, C.POINTER(C.c_uint32)]
    L.gsp_upload_accum.argtypes = [vp, vp, u64]
    L.gsp_get_stats.argtypes = [vp, C.POINTER(abi.Stats)]
    L.gsp_reset_stats.argtypes = [vp]
    L.gsp_trace.argtypes = [vp, vp, u64, C.c_int, vp]
    L.gsp_debug_visit_histograms.argtypes = [vp, vp, u64, vp, u64]
    L.gsp_last_error.argtypes = [vp]
    L.gsp_last_error.restype = C.c_char_p
    L.gsp_tile_partition.argtypes = [u32, u32, u32, u32, u32, vp]
    L.gsp_tile_partition.restype = u64
    L.gsp_multi_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.gsp_multi_create_ex.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(abi.CtxOptions), C.POINTER(vp)]
    L.gsp_multi_update_camera.argtypes = [vp, C.POINTER(abi.Camera)]
    L.gsp_multi_update_instances.argtypes = [vp, vp, u32]
    L.gsp_multi_update_tables.argtypes = [vp, C.POINTER(abi.SceneDesc)]
    L.gsp_multi_destroy.argtypes = [vp]
    L.gsp_multi_destroy.restype = None
    L.gsp_multi_num_shares.argtypes = [vp]
    L.gsp_multi_upload_scene.argtypes = [vp, C.POINTER(abi.SceneDesc)]
    L.gsp_multi_frame_begin.argtypes = [vp, u32, u32]
    L.gsp_multi_render.argtypes = [vp, C.POINTER(abi.RenderParams)]
    L.gsp_multi_sync.argtypes = [vp]
    L.gsp_multi_gather.argtypes = [vp, C.POINTER(vp)]
    L.gsp_multi_download.argtypes = [vp, vp]
    L.gsp_multi_get_stats.argtypes = [vp, C.POINTER(abi.Stats), C.POINTER(abi.Stats)]
    L.gsp_multi_reset_stats.argtypes = [vp]
    L.gsp_multi_gather_route.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    L.gsp_multi_last_error.argtypes = [vp]
    L.gsp_multi_last_error.restype = C.c_char_p
    v = L.gsp_abi_version()
    # (exact match: an older library fills arrays of gsp_stats at ITS struct size and knows nothing of gsp_render_params.disable_nee)
    if v != abi.GSP_ABI_VERSION:
        raise GspError("ABI version mismatch: abi.py is %d, %s is %d" % (abi.GSP_ABI_VERSION, path, v))
    _LIB = L
    return L


def device_count():
    return load().gsp_device_count()


def build_info():
    """What the loaded library was built from (gsp_build_info): {'arch', 'digest', 'flags'}."""
    text = load().gsp_build_info().decode()
    head, _, flags = text.partition(" flags=")
    info = dict(kv.split("=", 1) for kv in head.split())
    info["flags"] = flags
    return info


# the files csrc/Makefile hashes into the digest, in its order
DIGEST_SOURCES = ("pt_render.hip", "pt_bvh.hip", "pt_multi.hip", "pt_render_kernels.inc", "pt_render_scene.inc", "pt_render_pipeline.inc", "pt_wavetrace.h", "pt_versions.h", "pt_hostmath.h", "pt_math.h", "pt_shading.h",
                  "pt_trace.h", "pt_stages.h", "pt_internal.h", "../../include/gpuspectral_pt.h")


def source_digest():
    """The digest csrc/Makefile would stamp into a library built from the tree as it is now."""
    import hashlib

    h = hashlib.sha256()
    for f in DIGEST_SOURCES:
        with open(os.path.join(_HERE, "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


HIT_DT = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim", "<i4")])


def _options(options, kw):
    """abi.CtxOptions from an explicit struct, keyword fields, or -- test / A-B tooling only -- the GSP_* environment."""
    if options is None:
        options = abi.options_from_env()
    for k, v in kw.items():
        setattr(options, k, v)
    return options


class Context:
    """One HIP device + stream + scene + accumulate buffer (gsp_context).

    options: abi.CtxOptions (gsp_ctx_options); keyword arguments set single fields (pool_paths=..., primary_memo=2, ...).
    Without either, this TEST binding maps the GSP_* variables of the A/B scripts onto the struct (abi.options_from_env);
    the library itself never reads the environment."""

    def __init__(self, device=0, options=None, **option_fields):
        self._L = load()
        h = C.c_void_p()
        self.options = _options(options, option_fields)
        rc = self._L.gsp_ctx_create_ex(device, C.byref(self.options), C.byref(h))
        if rc != 0:
            raise GspError("gsp_ctx_create_ex: %s" % self._L.gsp_last_error(None).decode())
        self._h = h
        self._scene = None
        self.width = self.height = 0
        self.num_pixels = 0

    def _check(self, rc, what):
        if rc != 0:
            raise GspError("%s failed (%d): %s" % (what, rc, self._L.gsp_last_error(self._h).decode()))

    def close(self):
        if getattr(self, "_h", None):
            self._L.gsp_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upload_scene(self, scene):
        """scene: abi.SceneArrays."""
        d = scene.desc()
        self._scene = scene
        self._check(self._L.gsp_upload_scene(self._h, C.byref(d)), "gsp_upload_scene")

    def update_camera(self, to_world, fov):
        """gsp_update_camera: to_world = 16 floats in glm memory order, fov in radians."""
        cam = abi.Camera()
        for i, v in enumerate(np.asarray(to_world, np.float32).reshape(16)):
            cam.to_world[i] = float(v)
        cam.fov = float(fov)
        self._check(self._L.gsp_update_camera(self._h, C.byref(cam)), "gsp_update_camera")

    def update_instances(self, instances):
        """gsp_update_instances: abi.INSTANCE_DT records (same count and vertex ranges as the uploaded scene)."""
        inst = np.ascontiguousarray(instances, abi.INSTANCE_DT)
        self._check(self._L.gsp_update_instances(self._h, inst.ctypes.data if len(inst) else None, len(inst)), "gsp_update_instances")

    def update_tables(self, scene):
        """gsp_update_tables: the BSDF arrays and lights of `scene` (abi.SceneArrays) replace the resident ones."""
        d = scene.desc()
        self._check(self._L.gsp_update_tables(self._h, C.byref(d)), "gsp_update_tables")

    def frame_begin(self, width, height, pixel_ids=None):
        if pixel_ids is not None:
            pixel_ids = np.ascontiguousarray(pixel_ids, np.uint32)
            rc = self._L.gsp_frame_begin(self._h, width, height, pixel_ids.ctypes.data, len(pixel_ids))
            self.num_pixels = len(pixel_ids)
        else:
            rc = self._L.gsp_frame_begin(self._h, width, height, None, 0)
            self.num_pixels = width * height
        self._check(rc, "gsp_frame_begin")
        self.width, self.height = width, height

    def render(self, spp=1, first_timestamp=0, params=None, **overrides):
        p = params or abi.default_render_params()
        p.spp, p.first_timestamp = spp, first_timestamp
        for k, v in overrides.items():
            setattr(p, k, v)
        self._check(self._L.gsp_render(self._h, C.byref(p)), "gsp_render")

    def sync(self):
        self._check(self._L.gsp_sync(self._h), "gsp_sync")

    def peek(self):
        """(compact RGBA32F frame as it stands, timestamps folded into every pixel) without draining the pipeline."""
        out = np.zeros((self.num_pixels, 4), np.float32)
        folded = C.c_uint32(0)
        self._check(self._L.gsp_peek(self._h, out.ctypes.data, C.byref(folded)), "gsp_peek")
        return out, int(folded.value)

    def download(self, out=None):
        """The frame as (height, width, 4) float32; `out`: the caller's own framebuffer (C-contiguous float32 of that size)."""
        if out is None:
            out = np.zeros((self.height, self.width, 4), np.float32)
        assert out.dtype == np.float32 and out.size == self.height * self.width * 4 and out.flags.c_contiguous
        self._check(self._L.gsp_download(self._h, out.ctypes.data), "gsp_download")
        return out

    def download_compact(self):
        out = np.zeros((self.num_pixels, 4), np.float32)
        self._check(self._L.gsp_download_compact(self._h, out.ctypes.data), "gsp_download_compact")
        return out

    def copy_accum_to_device(self, device_ptr, nbytes):
        self._check(self._L.gsp_copy_accum_to_device(self._h, device_ptr, nbytes), "gsp_copy_accum_to_device")

    def peek_to_device(self, device_ptr, nbytes):
        """gsp_peek into device memory (e.g. a torch tensor's data_ptr()); returns the timestamps folded into every pixel."""
        folded = C.c_uint32(0)
        self._check(self._L.gsp_peek_to_device(self._h, device_ptr, nbytes, C.byref(folded)), "gsp_peek_to_device")
        return int(folded.value)

    def upload_accum(self, rgba):
        rgba = np.ascontiguousarray(rgba, np.float32).reshape(-1, 4)
        self._check(self._L.gsp_upload_accum(self._h, rgba.ctypes.data, len(rgba)), "gsp_upload_accum")

    def stats(self):
        s = abi.Stats()
        self._check(self._L.gsp_get_stats(self._h, C.byref(s)), "gsp_get_stats")
        return s.as_dict()

    def reset_stats(self):
        self._check(self._L.gsp_reset_stats(self._h), "gsp_reset_stats")

    def visit_histograms(self):
        """(visits per node index, tests per triangle slot) of the closest-hit rays traced with collect_traversal_stats=2."""
        st = self.stats()
        nodes = np.zeros(int(st["num_bvh_nodes"]), np.uint32)
        slots = np.zeros(int(st["num_triangles"]) + 8, np.uint32)
        self._check(self._L.gsp_debug_visit_histograms(self._h, nodes.ctypes.data, len(nodes), slots.ctypes.data, len(slots)),
                    "gsp_debug_visit_histograms")
        return nodes, slots

    def trace(self, rays, any_hit=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros(len(rays), HIT_DT)
        self._check(self._L.gsp_trace(self._h, rays.ctypes.data, len(rays), 1 if any_hit else 0, hits.ctypes.data),
                    "gsp_trace")
        return hits


def tile_partition(width, height, rank, world, tile=32):
    """Pixel ids of share `rank` of `world` (gsp_tile_partition: the C++ partition every multi-GPU path uses)."""
    L = load()
    n = L.gsp_tile_partition(width, height, rank, world, tile, None)
    ids = np.empty(n, np.uint32)
    if n:
        L.gsp_tile_partition(width, height, rank, world, tile, ids.ctypes.data)
    return ids


class MultiContext:
    """One frame over several GPUs of one node in ONE process (gsp_multi): tile partition, one host thread per share,
    device-to-device gather into the first device.  `devices` may repeat an index (several shares on one GPU)."""

    def __init__(self, devices, options=None, **option_fields):
        self._L = load()
        devs = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        self.options = _options(options, option_fields)
        rc = self._L.gsp_multi_create_ex(devs, len(devices), C.byref(self.options), C.byref(h))
        if rc != 0:
            raise GspError("gsp_multi_create_ex: %s" % self._L.gsp_multi_last_error(None).decode())
        self._h = h
        self._scene = None
        self.width = self.height = 0
        self.num_shares = len(devices)

    def _check(self, rc, what):
        if rc != 0:
            raise GspError("%s failed (%d): %s" % (what, rc, self._L.gsp_multi_last_error(self._h).decode()))

    def close(self):
        if getattr(self, "_h", None):
            self._L.gsp_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def upload_scene(self, scene):
        d = scene.desc()
        self._scene = scene
        self._check(self._L.gsp_multi_upload_scene(self._h, C.byref(d)), "gsp_multi_upload_scene")

    def update_camera(self, to_world, fov):
        cam = abi.Camera()
        for i, v in enumerate(np.asarray(to_world, np.float32).reshape(16)):
            cam.to_world[i] = float(v)
        cam.fov = float(fov)
        self._check(self._L.gsp_multi_update_camera(self._h, C.byref(cam)), "gsp_multi_update_camera")

    def update_instances(self, instances):
        inst = np.ascontiguousarray(instances, abi.INSTANCE_DT)
        self._check(self._L.gsp_multi_update_instances(self._h, inst.ctypes.data if len(inst) else None, len(inst)),
                    "gsp_multi_update_instances")

    def update_tables(self, scene):
        d = scene.desc()
        self._check(self._L.gsp_multi_update_tables(self._h, C.byref(d)), "gsp_multi_update_tables")

    def frame_begin(self, width, height):
        self._check(self._L.gsp_multi_frame_begin(self._h, width, height), "gsp_multi_frame_begin")
        self.width, self.height = width, height

    def render(self, spp=1, first_timestamp=0, params=None, **overrides):
        p = params or abi.default_render_params()
        p.spp, p.first_timestamp = spp, first_timestamp
        for k, v in overrides.items():
            setattr(p, k, v)
        self._check(self._L.gsp_multi_render(self._h, C.byref(p)), "gsp_multi_render")

    def sync(self):
        self._check(self._L.gsp_multi_sync(self._h), "gsp_multi_sync")

    def gather(self):
        """Assemble the frame on the first device; returns its device pointer (0 for a single share)."""
        p = C.c_void_p()
        self._check(self._L.gsp_multi_gather(self._h, C.byref(p)), "gsp_multi_gather")
        return p.value or 0

    def download(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._check(self._L.gsp_multi_download(self._h, out.ctypes.data), "gsp_multi_download")
        return out

    def stats(self, per_share=False):
        tot = abi.Stats()
        each = (abi.Stats * self.num_shares)()
        self._check(self._L.gsp_multi_get_stats(self._h, C.byref(tot), each), "gsp_multi_get_stats")
        return (tot.as_dict(), [e.as_dict() for e in each]) if per_share else tot.as_dict()

    def gather_route(self):
        """("rccl" | "copy", gathers carried by RCCL so far, gathers carried by peer copies so far)."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        r = self._L.gsp_multi_gather_route(self._h, C.byref(a), C.byref(b))
        return ("rccl" if r == 1 else "copy"), a.value, b.value

    def reset_stats(self):
        self._check(self._L.gsp_multi_reset_stats(self._h), "gsp_multi_reset_stats")
