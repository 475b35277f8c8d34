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
