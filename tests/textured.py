"""Scenes for the dormant-feature extension (include/gpuspectral_pt.h: textures + environment map).

`decorate` turns any flattened scene into one that uses the extension: random per-vertex uv (beyond [0, 1], so the
repeat wrap is exercised), random RGBA8 textures of awkward sizes, every record that can carry a texture gets one (or
none) at random, and a random lat-long HDR environment under a rotated frame.  `open_scene` is a scene rays can leave.
"""
import math

import numpy as np

from gpuspectral_amd import abi, scenes


def srgb_table():
    """byte -> linear value (IEC 61966-2-1), float32: passed as DATA to both sides."""
    c = np.arange(256, dtype=np.float64) / 255.0
    lin = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    return lin.astype(np.float32)


def rotation(ax_deg, ay_deg):
    ax, ay = math.radians(ax_deg), math.radians(ay_deg)
    rx = np.array([[1, 0, 0], [0, math.cos(ax), -math.sin(ax)], [0, math.sin(ax), math.cos(ax)]])
    ry = np.array([[math.cos(ay), 0, math.sin(ay)], [0, 1, 0], [-math.sin(ay), 0, math.cos(ay)]])
    m = np.eye(4)
    m[:3, :3] = rx @ ry
    return m.T.reshape(16).astype(np.float32)  # glm memory order (column-major)


def decorate(sc, seed=1, textures=True, envmap=True, decode=None, tex_sizes=((7, 5), (64, 33), (1, 1), (128, 128))):
    rng = np.random.RandomState(seed)
    if textures:
        sc.uvs = rng.uniform(-2.0, 3.0, (len(sc.positions), 2)).astype(np.float32)
        for (w, h) in tex_sizes:
            sc.add_texture(rng.randint(0, 256, (h, w, 4)).astype(np.uint8))
        for name in ("diffuse", "rough_conductor", "rough_plastic"):
            recs = sc.bsdfs[abi.BSDF_NAMES.index(name)]
            if len(recs):
                recs["has_texture"] = rng.randint(0, len(tex_sizes) + 1, len(recs))
        sc.texel_decode = decode
    if envmap:
        h, w = 16, 32
        env = rng.uniform(0.0, 1.5, (h, w, 4)).astype(np.float32)
        env[rng.randint(0, h), rng.randint(0, w), :3] = 40.0  # a sun: brighter than the firefly cutoff at weight 1
        sc.env_texels = env
        sc.env_to_local = rotation(20.0 + seed, -35.0)
    return sc


def open_scene(res=24):
    """Ground plane, three spheres (diffuse / rough conductor / rough plastic), a glass ball and a small emitter under
    an open sky."""
    b = scenes.SceneBuilder()
    rect = b.add_mesh(*scenes.rect_mesh())
    sph = b.add_mesh(*scenes.sphere_mesh(2 * res, res))
    b.add_object(rect, scenes.rowmajor([4, 0, 0, 0, 0, 0, 1, 0, 0, -4, 0, 0, 0, 0, 0, 1]), b.diffuse((0.5, 0.5, 0.5)))
    b.add_object(sph, scenes.trs((-0.9, 0.4, 0.0), 0.4), b.diffuse((0.7, 0.3, 0.3)))
    b.add_object(sph, scenes.trs((0.0, 0.4, 0.2), 0.4), b.rough_conductor(scenes.GOLD_ETA, scenes.GOLD_K, 0.15))
    b.add_object(sph, scenes.trs((0.9, 0.4, 0.0), 0.4), b.rough_plastic((0.2, 0.5, 0.8), 0.1, 1.4))
    b.add_object(sph, scenes.trs((0.3, 0.25, 0.9), 0.25), b.dielectric(1.5, 1.0), twofaced=False)
    b.add_object(rect, scenes.rowmajor([0.3, 0, 0, 0, 0, 0, -1, 2.0, 0, 0.3, 0, 0, 0, 0, 0, 1]), b.diffuse((0, 0, 0)),
                 emission=(30, 28, 25))
    b.camera_lookat((0.0, 1.2, 3.6), (0.0, 0.35, 0.0), fov_deg=38.0)
    return b.build()


# ---- a Mitsuba scene file that uses the loader's dormant features ----------------------------------------------------
DORMANT_XML = """<?xml version="1.0" encoding="utf-8"?>
<!-- build-authored test scene: bitmap / checkerboard textures on the three BSDFs that carry hasTexture, and an envmap
     emitter (the branches the reference keeps commented out, Loader.cpp:122-143,338-346) -->
<scene version="0.5.0">
  <sensor type="perspective">
    <float name="fov" value="38"/>
    <transform name="toWorld"><matrix value="-1 0 0 0 0 1 0 1.1 0 0 -1 4.2 0 0 0 1"/></transform>
  </sensor>
  <bsdf type="twosided" id="Ground"><bsdf type="diffuse">
    <texture name="reflectance" type="checkerboard">
      <float name="uscale" value="2"/><float name="vscale" value="3"/>
      <rgb name="color0" value="0.8, 0.75, 0.7"/><rgb name="color1" value="0.1, 0.15, 0.3"/>
    </texture></bsdf></bsdf>
  <bsdf type="twosided" id="Painted"><bsdf type="diffuse">
    <rgb name="reflectance" value="0.5, 0.5, 0.5"/>
    <texture name="reflectance" type="bitmap"><string name="filename" value="tex/paint.png"/></texture></bsdf></bsdf>
  <bsdf type="twosided" id="Wood"><bsdf type="roughplastic">
    <float name="alpha" value="0.1"/><float name="intIOR" value="1.5"/>
    <texture name="diffuseReflectance" type="bitmap"><string name="filename" value="tex/wood.jpg"/></texture></bsdf></bsdf>
  <bsdf type="twosided" id="Foil"><bsdf type="roughconductor">
    <float name="alpha" value="0.12"/><rgb name="eta" value="0.2, 0.9, 1.1"/><rgb name="k" value="3.9, 2.4, 2.2"/>
    <texture name="specularReflectance" type="bitmap"><string name="filename" value="tex/paint.png"/></texture></bsdf></bsdf>
  <bsdf type="twosided" id="Plain"><bsdf type="plastic">
    <float name="intIOR" value="1.4"/>
    <texture name="diffuseReflectance" type="bitmap"><string name="filename" value="tex/wood.jpg"/></texture></bsdf></bsdf>
  <shape type="rectangle"><transform name="toWorld"><matrix value="4 0 0 0 0 0 1 0 0 -4 0 0 0 0 0 1"/></transform><ref id="Ground"/></shape>
  <shape type="obj"><string name="filename" value="sphere.obj"/>
    <transform name="toWorld"><matrix value="0.45 0 0 -1.1 0 0.45 0 0.45 0 0 0.45 0 0 0 0 1"/></transform><ref id="Painted"/></shape>
  <shape type="obj"><string name="filename" value="sphere.obj"/>
    <transform name="toWorld"><matrix value="0.45 0 0 0 0 0.45 0 0.45 0 0 0.45 0.3 0 0 0 1"/></transform><ref id="Wood"/></shape>
  <shape type="obj"><string name="filename" value="sphere.obj"/>
    <transform name="toWorld"><matrix value="0.45 0 0 1.1 0 0.45 0 0.45 0 0 0.45 0 0 0 0 1"/></transform><ref id="Foil"/></shape>
  <shape type="cube"><transform name="toWorld"><matrix value="0.25 0 0 0.5 0 0.25 0 0.25 0 0 0.25 1.3 0 0 0 1"/></transform><ref id="Plain"/></shape>
  <shape type="rectangle">
    <transform name="toWorld"><matrix value="0.3 0 0 0 0 0 -1 2.2 0 0.3 0 0 0 0 0 1"/></transform>
    <bsdf type="diffuse"><rgb name="reflectance" value="0,0,0"/></bsdf>
    <emitter type="area"><rgb name="radiance" value="25, 24, 22"/></emitter>
  </shape>
  <emitter type="envmap"><string name="filename" value="tex/sky.pfm"/>
    <transform name="toWorld"><matrix value="0.8 0 0.6 0 0 1 0 0 -0.6 0 0.8 0 0 0 0 1"/></transform></emitter>
</scene>
"""


def write_dormant_scene(dirpath):
    """DORMANT_XML + its assets (sphere with uv, a PNG and a JPEG texture, a PFM sky) under `dirpath`; returns the xml path."""
    import os

    from PIL import Image

    os.makedirs(os.path.join(dirpath, "tex"), exist_ok=True)
    nu, nv = 24, 12  # lat-long sphere with vt
    with open(os.path.join(dirpath, "sphere.obj"), "w") as f:
        for j in range(nv + 1):
            for i in range(nu + 1):
                th, ph = math.pi * j / nv, 2 * math.pi * i / nu
                p = (math.sin(th) * math.cos(ph), math.cos(th), math.sin(th) * math.sin(ph))
                f.write("v %.9g %.9g %.9g\nvn %.9g %.9g %.9g\nvt %.9g %.9g\n" % (p + p + (2.0 * i / nu, 1.0 - j / nv)))
        for j in range(nv):
            for i in range(nu):
                a, b = j * (nu + 1) + i + 1, (j + 1) * (nu + 1) + i + 1
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a, a, a, a + 1, a + 1, a + 1, b, b, b))
                f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (a + 1, a + 1, a + 1, b + 1, b + 1, b + 1, b, b, b))
    rng = np.random.RandomState(4)
    y, x = np.mgrid[0:48, 0:80]
    paint = np.stack([128 + 100 * np.sin(x / 6.0), 128 + 100 * np.cos(y / 4.0), (x * 5 + y * 3) % 256], -1)
    Image.fromarray(np.clip(paint + rng.normal(0, 5, paint.shape), 0, 255).astype(np.uint8)).save(os.path.join(dirpath, "tex", "paint.png"))
    wood = np.stack([150 + 60 * np.sin(x / 3.0 + y / 11.0), 100 + 40 * np.sin(x / 3.0 + y / 11.0), 60 + 0 * x], -1)
    Image.fromarray(np.clip(wood, 0, 255).astype(np.uint8)).save(os.path.join(dirpath, "tex", "wood.jpg"), quality=90, subsampling=2)
    sky = np.zeros((16, 32, 3), np.float32)  # rows bottom-up in the file: row 0 = nadir
    sky[8:] = np.linspace(0.3, 1.2, 8)[:, None, None] * np.array([0.5, 0.7, 1.0], np.float32)
    sky[:8] = 0.1
    sky[12, 5] = 30.0
    with open(os.path.join(dirpath, "tex", "sky.pfm"), "wb") as f:
        f.write(b"PF\n32 16\n-1.0\n" + sky.astype("<f4").tobytes())
    xml = os.path.join(dirpath, "scene.xml")
    with open(xml, "w") as f:
        f.write(DORMANT_XML)
    return xml
