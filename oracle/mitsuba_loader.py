"""oracle/mitsuba_loader.py -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).

numpy restatement of the reference's Mitsuba-XML + OBJ scene loader,
``S/engine/Loader.cpp:19-64,145-234,253-349`` (S/ = src/GPUSpectral/), used to
check the product's C++ loader (gpuspectral_amd/host/Loader.cpp) array by array.

The two libraries the reference loader calls are NOT in /root/reference (empty
submodules, R/.gitmodules:10-12,34-36), so their behaviour is inferred from the
call sites -- PARITY UNPINNED for these points:
  * TinyParser-Mitsuba: camelCase property names are normalised to snake_case
    (toWorld -> to_world, intIOR -> int_ior); ``<ref id>`` resolves to the
    top-level object and appears among the anonymous children; a missing or
    wrongly-typed property yields the getter's default (number 0, colour 0,0,0,
    bool given default); ``<rgb value>`` takes 1 or 3 comma/space separated numbers.
  * tinyobjloader: faces with more than 3 corners are fan-triangulated
    (0, k-1, k); negative indices are relative to the current count.
"""
import math
import os
import re
import xml.etree.ElementTree as ET

import numpy as np

from gpuspectral_amd import abi

F = np.float32

# S/assets/rect.obj and S/assets/box.obj restated from their definition
# (SURVEY.md Appendix C): unit rectangle in z=0 with normal +z, faces
# "1 3 2 / 3 4 2"; cube of half-extent 1 with the same winding per face.
_RECT_V = [(-1.0, 1.0, 0.0), (1.0, 1.0, 0.0), (-1.0, -1.0, 0.0), (1.0, -1.0, 0.0)]
_BOX_FACES = [
    # (normal, four corners in rect.obj order)
    ((1, 0, 0), [(1, 1, 1), (1, 1, -1), (1, -1, 1), (1, -1, -1)]),
    ((-1, 0, 0), [(-1, 1, -1), (-1, 1, 1), (-1, -1, -1), (-1, -1, 1)]),
    ((0, 1, 0), [(-1, 1, -1), (1, 1, -1), (-1, 1, 1), (1, 1, 1)]),
    ((0, -1, 0), [(-1, -1, 1), (1, -1, 1), (-1, -1, -1), (1, -1, -1)]),
    ((0, 0, 1), [(-1, 1, 1), (1, 1, 1), (-1, -1, 1), (1, -1, 1)]),
    ((0, 0, -1), [(1, 1, -1), (-1, 1, -1), (1, -1, -1), (-1, -1, -1)]),
]


_QUAD_UV = [(0.0, 1.0), (1.0, 1.0), (0.0, 0.0), (1.0, 0.0)]  # vt of rect.obj, per corner


def _quad(corners, normal):
    order = (0, 2, 1, 2, 3, 1)  # f 1 3 2 / f 3 4 2
    pos = np.array([corners[i] for i in order], F)
    nrm = np.tile(np.array(normal, F), (6, 1))
    uv = np.array([_QUAD_UV[i] for i in order], F)
    return pos, nrm, uv


def rect_mesh(with_uv=False):
    pos, nrm, uv = _quad(_RECT_V, (0.0, 0.0, 1.0))
    return (pos, nrm, uv) if with_uv else (pos, nrm)


def box_mesh(with_uv=False):
    ps, ns, us = zip(*[_quad(c, n) for n, c in _BOX_FACES])
    if with_uv:
        return np.concatenate(ps), np.concatenate(ns), np.concatenate(us)
    return np.concatenate(ps), np.concatenate(ns)


def load_obj(path, with_uv=False):
    """Loader.cpp:19-64: de-indexed position/normal (and uv) arrays, all shapes concatenated."""
    v, vn, vt = [], [], []
    pos_idx, nrm_idx, uv_idx = [], [], []
    with open(path, "r", errors="replace") as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                v.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("vn "):
                p = line.split()
                vn.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("vt "):
                p = line.split()
                vt.append((float(p[1]), float(p[2])))
            elif line.startswith("f "):
                corners = []
                for tok in line.split()[1:]:
                    parts = tok.split("/")
                    vi = int(parts[0])
                    vi = vi - 1 if vi > 0 else len(v) + vi
                    ni = -1
                    if len(parts) >= 3 and parts[2]:
                        ni = int(parts[2])
                        ni = ni - 1 if ni > 0 else len(vn) + ni
                    ti = -1
                    if len(parts) >= 2 and parts[1]:
                        ti = int(parts[1])
                        ti = ti - 1 if ti > 0 else len(vt) + ti
                    corners.append((vi, ni, ti))
                for k in range(2, len(corners)):
                    for c in (corners[0], corners[k - 1], corners[k]):
                        pos_idx.append(c[0])
                        nrm_idx.append(c[1])
                        uv_idx.append(c[2])
    va = np.array(v, F).reshape(-1, 3)
    pos = va[np.array(pos_idx, np.int64)] if pos_idx else np.zeros((0, 3), F)
    ni = np.array(nrm_idx, np.int64)
    if len(vn) and (ni >= 0).all():
        nrm = np.array(vn, F).reshape(-1, 3)[ni]
    else:
        # the reference indexes attrib.normals unchecked (Loader.cpp:56-59); files
        # without vn are outside its contract.  Flat face normals are substituted.
        tri = pos.reshape(-1, 3, 3)
        fn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]).astype(F)
        ln = np.sqrt((fn * fn).sum(1, keepdims=True)).astype(F)
        fn = np.where(ln > 0, fn / np.where(ln > 0, ln, 1), 0).astype(F)
        nrm = np.repeat(fn, 3, axis=0)
    if with_uv:  # a corner without vt keeps Vertex::uv's default (0, 0)
        ti = np.array(uv_idx, np.int64)
        vta = np.concatenate([np.array(vt, F).reshape(-1, 2), np.zeros((1, 2), F)])
        uv = vta[np.where((ti >= 0) & (ti < len(vt)), ti, len(vt))] if len(ti) else np.zeros((0, 2), F)
        return np.ascontiguousarray(pos, F), np.ascontiguousarray(nrm, F), np.ascontiguousarray(uv, F)
    return np.ascontiguousarray(pos, F), np.ascontiguousarray(nrm, F)


def snake(name):
    """toWorld -> to_world, intIOR -> int_ior."""
    return re.sub(r"(?<=[a-z0-9])(?=[A-Z])", "_", name).lower()


class Obj:
    """Minimal stand-in for tinyparser_mitsuba::Object."""

    def __init__(self, kind, plugin):
        self.kind = kind  # 'shape' | 'bsdf' | 'sensor' | 'emitter' | other tag
        self.plugin = plugin
        self.props = {}  # snake name -> (type, value)
        self.children = []  # anonymous children (Obj)
        self.named = []  # (name, Obj)

    def number(self, name, default=0.0):
        p = self.props.get(name)
        return F(p[1]) if p and p[0] == "number" else F(default)

    def has(self, name):
        return name in self.props

    def color(self, name):
        p = self.props.get(name)
        if p and p[0] == "color":
            return np.array(p[1], F)
        return np.zeros(3, F)

    def string(self, name, default=""):
        p = self.props.get(name)
        return p[1] if p and p[0] == "string" else default

    def boolean(self, name, default=False):
        p = self.props.get(name)
        return p[1] if p and p[0] == "bool" else default


_OBJECT_TAGS = {"shape", "bsdf", "sensor", "emitter", "texture", "medium", "integrator", "sampler", "film", "rfilter"}


def _floats(text):
    return [float(t) for t in re.split(r"[\s,]+", text.strip()) if t]


def _parse_object(el, ids, pending=None):
    """`pending` collects forward <ref>s as (parent, slot, id); _resolve_refs patches them after the whole
    document has been read (same two passes as the C++ loader, gpuspectral_amd/host/Loader.cpp)."""
    o = Obj(el.tag, el.get("type", ""))
    for ch in el:
        tag = ch.tag
        name = snake(ch.get("name", ""))
        if tag in ("float", "integer"):
            o.props[name] = ("number", float(ch.get("value")))
        elif tag == "boolean":
            o.props[name] = ("bool", ch.get("value").strip().lower() == "true")
        elif tag == "string":
            o.props[name] = ("string", ch.get("value"))
        elif tag in ("rgb", "srgb"):
            vals = _floats(ch.get("value"))
            if len(vals) == 1:
                vals = vals * 3
            o.props[name] = ("color", vals[:3])
        elif tag in ("point", "vector"):
            if ch.get("value") is not None:
                vals = _floats(ch.get("value"))
            else:
                vals = [float(ch.get(k, "0")) for k in ("x", "y", "z")]
            o.props[name] = ("vector", vals[:3])
        elif tag == "transform":
            m = np.eye(4, dtype=np.float64)
            for t in ch:
                if t.tag == "matrix":
                    vals = _floats(t.get("value"))
                    m = np.array(vals, np.float64).reshape(4, 4)
            o.props[name] = ("transform", m.astype(F).reshape(16))  # 16 floats, row-major
        elif tag == "ref":
            target = ids.get(ch.get("id"))
            if target is not None:
                o.children.append(target)
            elif pending is not None:
                pending.append((o, len(o.children), ch.get("id")))
                o.children.append(None)
        elif tag in _OBJECT_TAGS:
            child = _parse_object(ch, ids, pending)
            if ch.get("id"):
                ids[ch.get("id")] = child
            if ch.get("name"):
                o.named.append((name, child))
            else:
                o.children.append(child)
    return o


def _resolve_refs(ids, pending, warnings):
    for parent, slot, rid in pending:
        if rid in ids:
            parent.children[slot] = ids[rid]
        else:
            warnings.append('unresolved <ref id="%s"> dropped' % rid)
    for parent, _, _ in pending:
        parent.children[:] = [c for c in parent.children if c is not None]


def glm_mul_point(m, p):
    """glm::mat4 * vec4(p, 1) in glm's association: (m0*x + m1*y) + (m2*z + m3*w)."""
    m = m.reshape(4, 4)  # m[c] = column c (glm memory order)
    x, y, z = F(p[0]), F(p[1]), F(p[2])
    return ((m[0] * x + m[1] * y) + (m[2] * z + m[3] * F(1.0))).astype(F)


def srgb_decode_table():
    """byte -> linear (IEC 61966-2-1), the table flattenScene hands to the renderer."""
    c = np.arange(256, dtype=np.float64) / 255.0
    return np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4).astype(F)


def read_bitmap(path):
    """loadTexture (Loader.cpp:66-86): RGBA8 words, last image row first, A = 0xFF.  PIL stands in for stb_image: PNG is
    lossless, so this equals the product's own decoder exactly; JPEG decoders agree to a few levels only."""
    from PIL import Image

    rgb = np.asarray(Image.open(path).convert("RGB"))[::-1]
    rgba = np.dstack([rgb, np.full(rgb.shape[:2], 255, np.uint8)])
    return np.ascontiguousarray(rgba)


def read_pfm(path):
    """RGBA32F, rows bottom-up (PFM's own order), A = 1."""
    with open(path, "rb") as f:
        magic = f.readline().strip()
        w, h = (int(t) for t in f.readline().split())
        scale = float(f.readline())
        ch = 3 if magic == b"PF" else 1
        data = np.frombuffer(f.read(4 * w * h * ch), "<f4" if scale < 0 else ">f4").reshape(h, w, ch).astype(F)
    if ch == 1:
        data = np.repeat(data, 3, axis=2)
    return np.ascontiguousarray(np.dstack([data, np.ones((h, w, 1), F)]))


def checkerboard(usize, vsize, color0, color1):
    """Loader.cpp:127-139 (dormant): 2*usize x 2*vsize cells of 100 x 100 texels."""
    def byte(c):
        return np.floor(np.clip(np.asarray(c, F), F(0.0), F(1.0)) * F(255.0) + F(0.5)).astype(np.uint8)  # float32 throughout

    y, x = np.mgrid[0 : vsize * 200, 0 : usize * 200]
    off = ((x // 100 + y // 100) & 1).astype(bool)
    img = np.zeros((vsize * 200, usize * 200, 4), np.uint8)
    img[..., :3] = np.where(off[..., None], byte(color1), byte(color0))
    img[..., 3] = 255
    return img


class _Builder:
    def __init__(self, asset_dir, parent="", dormant=False):
        self.asset_dir = asset_dir
        self.parent = parent
        self.dormant = dormant
        self.uv = []
        self.texture_cache = {}
        self.sc = abi.SceneArrays()
        self.meshes = {}  # path -> (first_vertex, vertex_count)
        self.pos, self.nrm = [], []
        self.nverts = 0
        self.instances = []
        self.bsdfs = [[] for _ in range(abi.BSDF_TYPE_COUNT)]
        self.lights = []
        self.warnings = []

    def mesh(self, path):
        if path in self.meshes:
            return self.meshes[path]
        base = os.path.basename(path)
        if path == "<builtin disk>":
            pos, nrm, uv = builtin_disk()
        elif path == "<builtin sphere>":
            pos, nrm, uv = builtin_sphere()
        elif not os.path.exists(path) and base == "rect.obj":
            pos, nrm, uv = rect_mesh(True)
        elif not os.path.exists(path) and base == "box.obj":
            pos, nrm, uv = box_mesh(True)
        else:
            pos, nrm, uv = load_obj(path, True)
        rec = (self.nverts, len(pos))
        self.pos.append(pos)
        self.nrm.append(nrm)
        self.uv.append(uv)
        self.nverts += len(pos)
        self.meshes[path] = rec
        return rec

    def add_bsdf(self, btype, fields):
        rec = np.zeros(1, abi.BSDF_DTYPES[btype])
        for k, v in fields.items():
            rec[k] = v
        self.bsdfs[btype].append(rec)
        return abi.bsdf_handle(btype, len(self.bsdfs[btype]) - 1)

    # Loader.cpp:122-143 (dormant in the reference): texture child `slot` -> has_texture value
    def load_texture(self, obj, slot, what):
        for name, tex in obj.named:
            if name != slot or tex.kind != "texture":
                continue
            if not self.dormant:
                self.warnings.append("%s: textured %s unsupported (Loader.cpp:122-143), colour default used" % (what, slot))
                return 0
            if tex.plugin == "bitmap":
                f = os.path.join(self.parent, tex.string("filename"))
                if f not in self.texture_cache and not os.path.exists(f):
                    self.warnings.append("%s: bitmap '%s' not found, colour default used" % (what, f))
                    return 0
                if f not in self.texture_cache:
                    self.texture_cache[f] = self.sc.add_texture(read_bitmap(f))
                return self.texture_cache[f]
            if tex.plugin == "checkerboard":
                return self.sc.add_texture(checkerboard(int(tex.number("uscale", 1.0)), int(tex.number("vscale", 1.0)),
                                                        tex.color("color0"), tex.color("color1")))
            self.warnings.append("%s: texture type '%s' unsupported, colour default used" % (what, tex.plugin))
            return 0
        return 0

    # Loader.cpp:145-234
    def load_material(self, mat, obj):
        t = obj.plugin
        if t == "twosided":
            mat["twofaced"] = 1
        if t == "diffuse":
            mat["bsdf"] = self.add_bsdf(abi.BSDF_DIFFUSE, {"reflectance": obj.color("reflectance"),
                                                           "has_texture": self.load_texture(obj, "reflectance", "diffuse")})
        elif t == "roughplastic":
            diffuse = obj.color("diffuse_reflectance")
            alpha = obj.number("alpha")
            ior = obj.number("int_ior") if obj.has("int_ior") else F(1.3)
            r0 = (ior - F(1.0)) / (ior + F(1.0))
            r0 = F(r0 * r0)
            mat["bsdf"] = self.add_bsdf(
                abi.BSDF_ROUGH_PLASTIC,
                {"diffuse": diffuse, "ior_in": ior, "ior_out": F(1.0), "r0": r0, "alpha": F(F(math.sqrt(2.0)) * alpha),
                 "has_texture": self.load_texture(obj, "diffuse_reflectance", "roughplastic")},
            )
        elif t == "dielectric":
            mat["bsdf"] = self.add_bsdf(
                abi.BSDF_SMOOTH_DIELECTRIC, {"ior_in": obj.number("int_ior"), "ior_out": obj.number("ext_ior")}
            )
        elif t == "conductor":
            ior = obj.number("eta") if obj.has("eta") else F(0.0)
            mat["bsdf"] = self.add_bsdf(abi.BSDF_SMOOTH_CONDUCTOR, {"ior_in": ior, "ior_out": F(1.0)})
        elif t == "plastic":
            diffuse = obj.color("diffuse_reflectance")
            ior = obj.number("int_ior") if obj.has("int_ior") else F(1.3)
            r0 = (ior - F(1.0)) / (ior + F(1.0))
            r0 = F(r0 * r0)
            if any(n == "diffuse_reflectance" for n, _ in obj.named):
                self.warnings.append("plastic: textured diffuse_reflectance unsupported (no hasTexture field), colour default used")
            mat["bsdf"] = self.add_bsdf(
                abi.BSDF_SMOOTH_PLASTIC, {"diffuse": diffuse, "ior_in": ior, "ior_out": F(1.0), "r0": r0}
            )
        elif t == "roughconductor":
            mat["bsdf"] = self.add_bsdf(
                abi.BSDF_ROUGH_CONDUCTOR,
                {
                    "eta": obj.color("eta"),
                    "k": obj.color("k"),
                    "reflectance": obj.color("specular_reflectance"),
                    "alpha": F(F(math.sqrt(2.0)) * obj.number("alpha")),
                    "has_texture": self.load_texture(obj, "specular_reflectance", "roughconductor"),
                },
            )
        for child in obj.children:
            if child.kind == "bsdf":
                self.load_material(mat, child)


def _unit_mid(a, b):
    """normalised midpoint in float32: + * sqrt / only, correctly rounded everywhere (host/Loader.cpp unitMid)"""
    x, y, z = F(a[0] + b[0]), F(a[1] + b[1]), F(a[2] + b[2])
    l = np.sqrt(F(F(F(x * x) + F(y * y)) + F(z * z)))
    return (F(x / l), F(y / l), F(z / l))


def builtin_disk():
    """LoadOptions::builtinShapes: Mitsuba's unit disk (z = 0, radius 1, normal +z) as 64 fan triangles
    -> (positions, normals, uvs) of the de-indexed mesh, the floats of host/Loader.cpp builtinDiskMesh"""
    N = 64
    p = [None] * N
    p[0], p[N // 4], p[N // 2], p[3 * N // 4] = (F(1), F(0), F(0)), (F(0), F(1), F(0)), (F(-1), F(0), F(0)), (F(0), F(-1), F(0))
    step = N // 4
    while step > 1:
        for i in range(0, N, step):
            p[i + step // 2] = _unit_mid(p[i], p[(i + step) % N])
        step //= 2
    pos = []
    for k in range(N):
        pos += [(F(0), F(0), F(0)), p[k], p[(k + 1) % N]]
    pos = np.array(pos, F)
    nrm = np.tile(np.array([0, 0, 1], F), (len(pos), 1))
    uv = (F(0.5) + F(0.5) * pos[:, :2]).astype(F)
    return pos, nrm, uv


def builtin_sphere():
    """LoadOptions::builtinShapes: the unit sphere as an octahedron subdivided three times (512 triangles), smooth normals
    -> (positions, normals, uvs), the floats of host/Loader.cpp builtinSphereMesh"""
    px, nx, py, ny, pz, nz = [(F(a), F(b), F(c)) for a, b, c in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1))]
    t = [(px, py, pz), (py, nx, pz), (nx, ny, pz), (ny, px, pz), (py, px, nz), (nx, py, nz), (ny, nx, nz), (px, ny, nz)]
    for _ in range(3):
        n = []
        for a, b, c in t:
            ab, bc, ca = _unit_mid(a, b), _unit_mid(b, c), _unit_mid(c, a)
            n += [(a, ab, ca), (ab, b, bc), (ca, bc, c), (ab, bc, ca)]
        t = n
    pos = np.array([q for f in t for q in f], F)
    uv = (F(0.5) + F(0.5) * pos[:, :2]).astype(F)
    return pos, pos.copy(), uv


def load_scene(path, asset_dir=None, dormant_features=False, srgb_textures=True, builtin_shapes=False):
    """Loader.cpp:253-349 -> abi.SceneArrays.  dormant_features: LoadOptions::dormantFeatures of the product's loader
    (the texture and envmap branches the reference keeps commented out, Loader.cpp:122-143,338-346); builtin_shapes:
    LoadOptions::builtinShapes (`disk` / `sphere` tessellated instead of skipped)."""
    parent = os.path.dirname(os.path.abspath(path))
    asset_dir = asset_dir or parent
    root = ET.parse(path).getroot()
    ids = {}
    pending = []
    top = _parse_object(root, ids, pending)
    b = _Builder(asset_dir, parent, dormant_features)
    _resolve_refs(ids, pending, b.warnings)
    sc = b.sc
    for obj in top.children:
        if obj.kind == "shape":
            pt = obj.plugin
            if pt == "obj":
                filename = os.path.join(parent, obj.string("filename"))
            elif pt == "rectangle":
                filename = os.path.join(asset_dir, "rect.obj")
            elif pt == "cube":
                filename = os.path.join(asset_dir, "box.obj")
            elif pt == "disk":
                filename = os.path.join(asset_dir, "disk.obj")
                if builtin_shapes and not os.path.exists(filename):
                    filename = "<builtin disk>"
            elif pt == "sphere" and builtin_shapes:
                filename = "<builtin sphere>"
            else:
                b.warnings.append("unsupported shape type '%s' skipped" % pt)
                continue
            if not (os.path.exists(filename) or filename.startswith("<builtin") or (pt != "obj" and os.path.basename(filename) in ("rect.obj", "box.obj"))):
                b.warnings.append("missing mesh '%s' skipped" % filename)
                continue
            first, count = b.mesh(filename)
            tr = obj.props.get("to_world")
            rowmajor = tr[1] if tr and tr[0] == "transform" else np.eye(4, dtype=F).reshape(16)
            matrix = rowmajor.reshape(4, 4).T.copy().reshape(16)  # glm::transpose(make_mat4(row-major))
            c = obj.props.get("center")
            if c and c[0] == "vector":  # Loader.cpp:288-293
                matrix[12:16] = np.array([c[1][0], c[1][1], c[1][2], 1.0], F)
            if filename == "<builtin sphere>":  # (extension: the reference never reads `radius`)
                r = obj.number("radius") if obj.has("radius") else F(1.0)
                m3 = matrix.reshape(4, 4)
                m3[:3, :3] = (m3[:3, :3] * F(r)).astype(F)
            mat = {"emission": np.zeros(3, F), "twofaced": 0, "bsdf": 0}
            emitting = False
            for child in obj.children:
                if child.kind == "bsdf":
                    b.load_material(mat, child)
                elif child.kind == "emitter" and child.plugin == "area":
                    mat["emission"] = child.color("radiance")
                    emitting = True
            inst = np.zeros(1, abi.INSTANCE_DT)
            inst["transform"] = matrix
            inst["emission"] = mat["emission"]
            inst["bsdf"] = mat["bsdf"]
            inst["twofaced"] = mat["twofaced"]
            inst["first_vertex"] = first
            inst["vertex_count"] = count
            b.instances.append(inst)
            if emitting:  # Loader.cpp:316-330
                pos = np.concatenate(b.pos)[first : first + count] if b.pos else np.zeros((0, 3), F)
                for i in range(0, count - 2, 3):
                    lt = np.zeros(1, abi.LIGHT_DT)
                    for k in range(3):
                        lt["positions"][0, k] = glm_mul_point(matrix, pos[i + k])
                    lt["radiance"][0, :3] = mat["emission"]
                    lt["radiance"][0, 3] = 1.0
                    b.lights.append(lt)
        elif obj.kind == "sensor":  # Loader.cpp:331-337
            tr = obj.props.get("to_world")
            rowmajor = tr[1] if tr and tr[0] == "transform" else np.eye(4, dtype=F).reshape(16)
            fov = obj.number("fov")
            sc.fov = F(np.float64(fov) * math.pi / np.float64(F(180.0)))
            sc.to_world = rowmajor.reshape(4, 4).T.copy().reshape(16)
        elif obj.kind == "emitter":
            if not dormant_features or obj.plugin != "envmap":
                b.warnings.append("top-level emitter (envmap) ignored, as in the reference (Loader.cpp:338-346)")
            else:  # Loader.cpp:339-345 (dormant)
                tr = obj.props.get("to_world")
                rowmajor = tr[1] if tr and tr[0] == "transform" else np.eye(4, dtype=F).reshape(16)
                sc.env_texels = read_pfm(os.path.join(parent, obj.string("filename")))
                to_world = rowmajor.reshape(4, 4).astype(np.float64)  # row-major = the mathematical matrix
                sc.env_to_local = np.linalg.inv(to_world).T.reshape(16).astype(F)  # glm memory order
    if len(sc.textures):
        sc.uvs = np.concatenate(b.uv) if b.uv else np.zeros((0, 2), F)
        sc.texel_decode = srgb_decode_table() if srgb_textures else (np.arange(256, dtype=F) / F(255.0)).astype(F)
    sc.instances = np.concatenate(b.instances) if b.instances else np.zeros(0, abi.INSTANCE_DT)
    sc.positions = np.concatenate(b.pos) if b.pos else np.zeros((0, 3), F)
    sc.normals = np.concatenate(b.nrm) if b.nrm else np.zeros((0, 3), F)
    sc.bsdfs = [
        np.concatenate(lst) if lst else np.zeros(0, dt) for lst, dt in zip(b.bsdfs, abi.BSDF_DTYPES)
    ]
    sc.lights = np.concatenate(b.lights) if b.lights else np.zeros(0, abi.LIGHT_DT)
    sc.warnings = b.warnings
    return sc
