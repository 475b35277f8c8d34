"""ctypes binding of libgpuspectral_host.so: the C++ host layer (Scene, loadScene, PathTracer).

The classes here only forward to the C++ objects a C++ caller would use
(gpuspectral_amd/host/*.h); see those headers for the reference citations.
"""
import ctypes as C
import os

import numpy as np

from . import abi
from .pt import GspError

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib_path():
    return os.path.join(_HERE, "lib", "libgpuspectral_host.so")


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise GspError("host library %s not found: build it with `make -C gpuspectral_amd/host`" % path)
    L = C.CDLL(path)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    L.gsph_last_error.restype = C.c_char_p
    L.gsph_load_scene.restype = vp
    L.gsph_load_scene.argtypes = [C.c_char_p, C.c_char_p]
    L.gsph_load_scene_ex.restype = vp
    L.gsph_load_scene_ex.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    L.gsph_load_scene_opts.restype = vp
    L.gsph_load_scene_opts.argtypes = [C.c_char_p, C.c_char_p, C.c_uint]
    L.gsph_load_bitmap.argtypes = [C.c_char_p, C.POINTER(u32), C.POINTER(u32), vp]
    L.gsph_load_hdr_bitmap.argtypes = [C.c_char_p, C.POINTER(u32), C.POINTER(u32), vp]
    L.gsph_scene_free.argtypes = [vp]
    L.gsph_scene_free.restype = None
    L.gsph_scene_desc.restype = C.POINTER(abi.SceneDesc)
    L.gsph_scene_desc.argtypes = [vp]
    L.gsph_scene_num_warnings.restype = u32
    L.gsph_scene_num_warnings.argtypes = [vp]
    L.gsph_scene_warning.restype = C.c_char_p
    L.gsph_scene_warning.argtypes = [vp, u32]
    L.gsph_scene_num_materials.restype = u32
    L.gsph_scene_num_materials.argtypes = [vp]
    L.gsph_scene_num_objects.restype = u32
    L.gsph_scene_num_objects.argtypes = [vp]
    L.gsph_scene_set_camera.argtypes = [vp, vp, C.c_float]
    L.gsph_scene_set_transform.argtypes = [vp, u32, vp]
    L.gsph_scene_set_object_material.argtypes = [vp, u32, vp, C.c_int]
    L.gsph_scene_set_diffuse_reflectance.argtypes = [vp, u32, vp]
    L.gsph_scene_make_object_rough_conductor.argtypes = [vp, u32, vp, vp, C.c_float]
    L.gsph_scene_reflatten.argtypes = [vp]
    L.gsph_pathtracer_render_files_same_slot.argtypes = [vp, C.POINTER(C.c_char_p), C.c_char_p, u32, u32, vp, vp]
    L.gsph_tracker_remember.argtypes = [vp]
    L.gsph_tracker_diff.argtypes = [vp]
    L.gsph_tracker_probe.argtypes = [vp, vp]
    L.gsph_pathtracer_create.restype = vp
    L.gsph_pathtracer_create.argtypes = [u32, u32, C.c_int, vp, u64]
    L.gsph_pathtracer_free.argtypes = [vp]
    L.gsph_pathtracer_free.restype = None
    L.gsph_pathtracer_create_render_pass.argtypes = [vp, vp]
    L.gsph_pathtracer_render.argtypes = [vp, vp, u32]
    L.gsph_pathtracer_set_params.argtypes = [vp, C.POINTER(abi.RenderParams)]
    L.gsph_pathtracer_timestamp.argtypes = [vp]
    L.gsph_pathtracer_reset.argtypes = [vp]
    L.gsph_pathtracer_download.argtypes = [vp, vp, u64]
    L.gsph_pathtracer_stats.argtypes = [vp, C.POINTER(abi.Stats)]
    L.gsph_write_pfm.argtypes = [C.c_char_p, vp, u32, u32]
    L.gsph_write_ppm.argtypes = [C.c_char_p, vp, u32, u32, C.c_int]
    L.gsph_tone_map.argtypes = [vp, u32, u32, C.c_int, vp]
    _LIB = L
    return L


def _err(L):
    return L.gsph_last_error().decode(errors="replace")


def _view(ptr, count, dtype):
    if not ptr or count == 0:
        return np.zeros(0, dtype)
    buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


class Scene:
    """A C++ GPUSpectral::Scene produced by loadScene (S/engine/Loader.cpp:253-349)."""

    def __init__(self, path, asset_dir=None, dormant_features=False, srgb_textures=True, builtin_shapes=False):
        """dormant_features: LoadOptions::dormantFeatures (textures / environment map); builtin_shapes:
        LoadOptions::builtinShapes (`disk` and `sphere` shapes are built instead of skipped); the defaults are
        the reference's behaviour."""
        self._L = load()
        ad = asset_dir.encode() if asset_dir else None
        if builtin_shapes:
            self._h = self._L.gsph_load_scene_opts(path.encode(), ad, (1 if dormant_features else 0) | (2 if srgb_textures else 0) | 4)
        elif dormant_features:
            self._h = self._L.gsph_load_scene_ex(path.encode(), ad, 1, 1 if srgb_textures else 0)
        else:
            self._h = self._L.gsph_load_scene(path.encode(), ad)
        if not self._h:
            raise GspError("loadScene(%s): %s" % (path, _err(self._L)))

    def close(self):
        if getattr(self, "_h", None):
            self._L.gsph_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def warnings(self):
        return [self._L.gsph_scene_warning(self._h, i).decode() for i in range(self._L.gsph_scene_num_warnings(self._h))]

    @property
    def num_materials(self):
        return self._L.gsph_scene_num_materials(self._h)

    # ---- edits between frames (a host application mutates its Scene; PathTracer::prepareScene compares by value) ----
    def _ok(self, rc, what):
        if rc != 0:
            raise GspError("%s: %s" % (what, _err(self._L)))

    @property
    def num_objects(self):
        return self._L.gsph_scene_num_objects(self._h)

    def set_camera(self, to_world, fov):
        m = np.ascontiguousarray(to_world, np.float32).reshape(16)
        self._ok(self._L.gsph_scene_set_camera(self._h, m.ctypes.data, float(fov)), "set_camera")

    def set_transform(self, obj, matrix16):
        m = np.ascontiguousarray(matrix16, np.float32).reshape(16)
        self._ok(self._L.gsph_scene_set_transform(self._h, obj, m.ctypes.data), "set_transform")

    def set_object_material(self, obj, emission=None, twofaced=None):
        e = np.ascontiguousarray(emission, np.float32).reshape(3) if emission is not None else None
        self._ok(self._L.gsph_scene_set_object_material(self._h, obj, e.ctypes.data if e is not None else None,
                                                        -1 if twofaced is None else int(bool(twofaced))), "set_object_material")

    def set_diffuse_reflectance(self, index, rgb):
        c = np.ascontiguousarray(rgb, np.float32).reshape(3)
        self._ok(self._L.gsph_scene_set_diffuse_reflectance(self._h, index, c.ctypes.data), "set_diffuse_reflectance")

    def make_object_rough_conductor(self, obj, eta, k, alpha):
        e, kk = np.ascontiguousarray(eta, np.float32).reshape(3), np.ascontiguousarray(k, np.float32).reshape(3)
        self._ok(self._L.gsph_scene_make_object_rough_conductor(self._h, obj, e.ctypes.data, kk.ctypes.data, float(alpha)),
                 "make_object_rough_conductor")

    def arrays(self):
        """Copy of the flattened scene AS IT IS NOW (what crosses the C ABI) as abi.SceneArrays."""
        self._ok(self._L.gsph_scene_reflatten(self._h), "flattenScene")
        d = self._L.gsph_scene_desc(self._h).contents
        sc = abi.SceneArrays()
        sc.instances = _view(d.instances, d.num_instances, abi.INSTANCE_DT)
        sc.positions = _view(d.positions, d.num_vertices * 3, np.float32).reshape(-1, 3)
        sc.normals = _view(d.normals, d.num_vertices * 3, np.float32).reshape(-1, 3)
        sc.bsdfs = [
            _view(getattr(d, name + "_bsdfs"), d.num_bsdfs[i], dt)
            for i, (name, dt) in enumerate(zip(abi.BSDF_NAMES, abi.BSDF_DTYPES))
        ]
        sc.lights = _view(d.lights, d.num_lights, abi.LIGHT_DT)
        sc.to_world = np.array(list(d.camera.to_world), np.float32)
        sc.fov = np.float32(d.camera.fov)
        if d.num_textures:
            sc.uvs = _view(d.uvs, d.num_vertices * 2, np.float32).reshape(-1, 2)
            sc.textures = _view(d.textures, d.num_textures, abi.TEXTURE_DT)
            sc.texels = _view(d.texels, d.num_texels, np.uint32)
            sc.texel_decode = _view(d.texel_decode, 256, np.float32) if d.texel_decode else None
        if d.envmap.texels:
            w, h = d.envmap.width, d.envmap.height
            sc.env_texels = _view(d.envmap.texels, w * h * 4, np.float32).reshape(h, w, 4)
            sc.env_to_local = np.array(list(d.envmap.to_local), np.float32)
        return sc


def load_bitmap(path):
    """Image.h loadBitmap: (H, W, 4) uint8, row 0 = bottom image row."""
    L = load()
    w, h = C.c_uint32(), C.c_uint32()
    if L.gsph_load_bitmap(path.encode(), C.byref(w), C.byref(h), None):
        raise GspError("loadBitmap(%s): %s" % (path, _err(L)))
    out = np.zeros((h.value, w.value), np.uint32)
    if L.gsph_load_bitmap(path.encode(), C.byref(w), C.byref(h), out.ctypes.data):
        raise GspError("loadBitmap(%s): %s" % (path, _err(L)))
    return out.view(np.uint8).reshape(h.value, w.value, 4)


def load_hdr_bitmap(path):
    """Image.h loadHdrBitmap: (H, W, 4) float32, row 0 = bottom image row."""
    L = load()
    w, h = C.c_uint32(), C.c_uint32()
    if L.gsph_load_hdr_bitmap(path.encode(), C.byref(w), C.byref(h), None):
        raise GspError("loadHdrBitmap(%s): %s" % (path, _err(L)))
    out = np.zeros((h.value, w.value, 4), np.float32)
    if L.gsph_load_hdr_bitmap(path.encode(), C.byref(w), C.byref(h), out.ctypes.data):
        raise GspError("loadHdrBitmap(%s): %s" % (path, _err(L)))
    return out


class PathTracer:
    """C++ GPUSpectral::PathTracer (drop-in for S/renderer/PathTracer.h:48-68)."""

    def __init__(self, width, height, device=0, pixel_ids=None):
        self._L = load()
        ids = np.ascontiguousarray(pixel_ids, np.uint32) if pixel_ids is not None else None
        self._h = self._L.gsph_pathtracer_create(width, height, device, ids.ctypes.data if ids is not None else None,
                                                 len(ids) if ids is not None else 0)
        if not self._h:
            raise GspError("PathTracer(): %s" % _err(self._L))
        self.width, self.height = width, height

    def _check(self, rc, what):
        if rc != 0:
            raise GspError("%s: %s" % (what, _err(self._L)))

    def close(self):
        if getattr(self, "_h", None):
            self._L.gsph_pathtracer_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def create_render_pass(self, scene):
        self._check(self._L.gsph_pathtracer_create_render_pass(self._h, scene._h), "createRenderPass")

    def render(self, scene, spp):
        self._check(self._L.gsph_pathtracer_render(self._h, scene._h, spp), "render")

    def render_files_same_slot(self, paths, spp, asset_dir=None):
        """Loads each file into a Scene in ONE stack slot (a C++ loop-local), renders `spp` samples from timestamp 0 and
        downloads: (images [n, H, W, 4], addresses of the n Scene objects)."""
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        out = np.zeros((len(paths), self.height, self.width, 4), np.float32)
        addr = np.zeros(len(paths), np.uint64)
        self._check(self._L.gsph_pathtracer_render_files_same_slot(self._h, arr, asset_dir.encode() if asset_dir else None,
                                                                   len(paths), spp, out.ctypes.data, addr.ctypes.data),
                    "render_files_same_slot")
        return out, addr

    def set_params(self, params):
        self._check(self._L.gsph_pathtracer_set_params(self._h, C.byref(params)), "set_params")

    @property
    def timestamp(self):
        return self._L.gsph_pathtracer_timestamp(self._h)

    def reset(self):
        self._check(self._L.gsph_pathtracer_reset(self._h), "reset")

    def download(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._check(self._L.gsph_pathtracer_download(self._h, out.ctypes.data, out.size), "download")
        return out

    def stats(self):
        s = abi.Stats()
        self._check(self._L.gsph_pathtracer_stats(self._h, C.byref(s)), "stats")
        return s.as_dict()


def write_pfm(path, rgba):
    rgba = np.ascontiguousarray(rgba, np.float32)
    h, w = rgba.shape[0], rgba.shape[1]
    L = load()
    if L.gsph_write_pfm(path.encode(), rgba.ctypes.data, w, h) != 0:
        raise GspError("write_pfm: %s" % _err(L))


def tone_map(rgba, aces=False):
    """uint8 RGB image: clamp (or ACESFilm, S/assets/shaders/common.glsl:74-82) then gamma 2.2."""
    rgba = np.ascontiguousarray(rgba, np.float32)
    h, w = rgba.shape[0], rgba.shape[1]
    out = np.zeros((h, w, 3), np.uint8)
    L = load()
    if L.gsph_tone_map(rgba.ctypes.data, w, h, 1 if aces else 0, out.ctypes.data) != 0:
        raise GspError("tone_map: %s" % _err(L))
    return out


def write_ppm(path, rgba, aces=False):
    rgba = np.ascontiguousarray(rgba, np.float32)
    L = load()
    if L.gsph_write_ppm(path.encode(), rgba.ctypes.data, rgba.shape[1], rgba.shape[0], 1 if aces else 0) != 0:
        raise GspError("write_ppm: %s" % _err(L))
