#!/usr/bin/env python3
"""Bias attribution for the CPU oracle (TEST INFRASTRUCTURE; CPU only, no GPU, no /root/reference).

    python tests/tools/quirk_probe.py [--size 512] [--spp 64] > profiles/r06_quirk_attribution.txt

Renders the Cornell box with the oracle under every setting of oracle_set_quirks_off (oracle/oracle_bsdf.h, kQuirk*)
and compares each render, cell by cell, with the reference-held Tungsten image (tests/golden/ref_scenes/
tungsten_cornell-box.npz; cells as in tests/test_gpu_reference_images.py: 8 x 8 cells of 16 x 16 fixture blocks).

Mask 0 is the reference's estimator (rayhit.rchit:751,763-790, raygen.rgen:60-63); mask 15 replaces all four documented
departures by their textbook forms.  If the restatement reads the shaders right, mask 15 must agree with an unbiased path
tracer everywhere, and the gap between mask 0 and Tungsten is the reference's own bias, quirk by quirk.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

QUIRKS = {1: "NEE weight with the sampled direction's pdf (rchit:751)",
          2: "emitter-hit weight from the previous light SAMPLE's pdf / 1 when shadowed (rchit:763-765,785-790)",
          4: "firefly cutoff 20 (rgen:60-63)",
          8: "rough-plastic eval pdf floor 0.01 (rchit:577)"}


def cells_of(img, fix, size):
    from test_gpu_reference_images import _cells

    blocks = fix["lin"].shape[0]
    k = size // blocks
    ours = img[:, :3].astype(np.float64).reshape(blocks, k, blocks, k, 3).mean(axis=(1, 3))
    return _cells(ours, fix, strict=True)[0]


def render(orc, scene, size, spp, mask):
    L = orc.lib()
    L.oracle_set_quirks_off(mask)
    try:
        o = orc.Oracle(scene)
        img, st = o.render(size, size, spp=spp)
        o.close()
    finally:
        L.oracle_set_quirks_off(0)
    return img, st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--masks", default="0,1,2,4,3,7,15")
    a = ap.parse_args()
    from conftest import CORNELL_XML, GOLDEN
    from oracle import mitsuba_loader as ml
    from oracle import oracle as orc

    scene = ml.load_scene(CORNELL_XML)
    fix = np.load(os.path.join(GOLDEN, "ref_scenes", "tungsten_cornell-box.npz"))
    print("# Cornell box, oracle %d x %d x %d spp against tungsten_cornell-box.npz; ratio = oracle / Tungsten per cell-channel" % (a.size, a.size, a.spp))
    for b, t in QUIRKS.items():
        print("#   bit %d: %s" % (b, t))
    base = None
    for mask in [int(m) for m in a.masks.split(",")]:
        t0 = time.time()
        img, st = render(orc, scene, a.size, a.spp, mask)
        r = cells_of(img, fix, a.size)
        v = r[np.isfinite(r)]
        walls = r[1:3, 1:7]
        print("\nquirks_off = %2d   cells %d   min %.3f  max %.3f  median %.3f  mean|r-1| %.4f   lit walls %.3f .. %.3f   (%.0f s, %d ext rays)"
              % (mask, v.size, v.min(), v.max(), np.median(v), np.abs(v - 1).mean(), np.nanmin(walls), np.nanmax(walls),
                 time.time() - t0, st["extension_rays"]))
        g = np.nanmean(r, axis=2)  # mean over channels
        for row in g:
            print("   " + " ".join("  -  " if not np.isfinite(x) else "%5.3f" % x for x in row))
        if mask == 0:
            base = r
        elif base is not None:
            d = (r / base)[np.isfinite(r) & np.isfinite(base)]
            print("   against quirks_off = 0: ratio min %.3f max %.3f median %.3f" % (d.min(), d.max(), np.median(d)))


if __name__ == "__main__":
    main()
