#!/usr/bin/env python3
"""Reference-held image fixtures: the third-party Tungsten ground-truth renders the reference ships beside its scenes
(S/assets/scenes/<scene>/TungstenRender.png) -> linear radiance, float32, box-filtered 8x8.

    python tests/golden/make_tungsten.py          # needs /root/reference (build container only)

They are the only expected OUTPUTS that exist in the reference tree (its .exr twins are HALF+PIZ, no decoder in this
image); the GPU tests compare the HIP renders of the same scenes with them region by region
(tests/test_gpu_reference_images.py).  Data only (images), no source.

Linearisation.  The PNGs were written by Tungsten with its `filmic` tone map (Hejl / Burgess-Dawson:
x = max(c - 0.004, 0); y = x (6.2 x + 0.5) / (x (6.2 x + 1.7) + 0.06), display gamma folded in), not with the plain
gamma 2.2 the Mitsuba XML's <film> block names.  Evidence, from the Cornell box (diffuse only, so the reference
integrator, this repository's renders and an unbiased path tracer must agree to within the reference's small MIS
bias): under gamma-2.2 linearisation the blue channel of every lit wall is 3x off and unlit regions 6-60x, with no
single exposure that fits; under the inverse filmic curve all three channels of all lit walls agree with the HIP
render to 1-5 % (profiles/r02_tungsten_compare.txt).  The curve is inverted in closed form below.

Per scene the file holds: `lin` [h/8, w/8, 3] linear radiance, `sat` [h/8, w/8, 3] = some pixel of the 8x8 block is
clipped (>= 254) or in the toe (<= 2) in that channel of the PNG, i.e. the block's radiance is not recoverable;
`shoulder` [h/8, w/8, 3] (r06) = some pixel of the block is >= 232 in that channel: on the curve's shoulder one 8-bit level is
> 5 % of radiance (17 % at 248, 51 % at 252 -- the Cornell light is written as 252 / 250 / 243 for radiance 17 / 12 / 4), so
such blocks pass the `sat` test but carry no usable number.  Tests that claim a few per cent exclude them too.
"""
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
SCENES = "/root/reference/src/GPUSpectral/assets/scenes"
BLOCK = 8


def inverse_filmic(y):
    y = np.clip(y, 0.0, 0.999)
    a = 6.2 - 6.2 * y
    b = 0.5 - 1.7 * y
    c = -0.06 * y
    x = (-b + np.sqrt(np.maximum(b * b - 4.0 * a * c, 0.0))) / (2.0 * a)
    return x + 0.004 * (y > 0)


def main():
    for name in ("cornell-box", "staircase2", "coffee"):
        raw = np.asarray(Image.open(os.path.join(SCENES, name, "TungstenRender.png")).convert("RGB"))
        lin = inverse_filmic(raw.astype(np.float64) / 255.0)
        h, w = lin.shape[:2]
        lin = lin.reshape(h // BLOCK, BLOCK, w // BLOCK, BLOCK, 3).mean(axis=(1, 3))
        bad = ((raw >= 254) | (raw <= 2)).reshape(h // BLOCK, BLOCK, w // BLOCK, BLOCK, 3).any(axis=(1, 3))
        out = os.path.join(HERE, "ref_scenes", "tungsten_%s.npz" % name)
        shoulder = (raw >= 232).reshape(h // BLOCK, BLOCK, w // BLOCK, BLOCK, 3).any(axis=(1, 3))
        np.savez_compressed(out, lin=lin.astype(np.float32), sat=bad, shoulder=shoulder)
        print(out, lin.shape, lin.mean((0, 1)), "unrecoverable blocks: %.1f %%" % (100.0 * bad.mean()))


if __name__ == "__main__":
    main()
