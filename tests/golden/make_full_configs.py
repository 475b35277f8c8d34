#!/usr/bin/env python3
"""Generates tests/golden/full_configs.npz: oracle expectations for the BASELINE.json configs at their FULL sizes.

For every config (`tests/full_configs.py::CONFIGS`) the scalar CPU oracle renders a fixed sparse pixel subset
(>= 256 pixels, seeded choice) of the full-resolution frame twice: at a short sample count (SPP_QUICK) and at the
config's full sample count.  The seed of a path is a function of (global pixel index, timestamp) only
(raygen.rgen:37), so the subset reproduces exactly those pixels of the full frame, and the GPU tests
(tests/test_gpu_full_configs.py) compare them bit for bit -- the frame itself is rendered at full resolution and
full spp on the GPU.  Stored per config: pixel ids, RGBA at both sample counts, and the oracle's ray / vertex
counts for the subset (the GPU must trace exactly as many).

Run here (needs only gcc + numpy; ~1-2 min on 8 cores):   python tests/golden/make_full_configs.py
The scenes come from the seeded procedural generators in gpuspectral_amd/scenes.py (no reference data).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from full_configs import CONFIGS, SPP_QUICK, config_params, config_pixels, config_scene  # noqa: E402
from oracle import oracle as orc  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "full_configs.npz")


def main():
    out = {}
    for name, cfg in CONFIGS.items():
        t0 = time.time()
        sc = config_scene(cfg)
        ids = config_pixels(cfg)
        o = orc.Oracle(sc)
        W, H = cfg["width"], cfg["height"]
        acc, st_q = o.render(W, H, spp=SPP_QUICK, pixel_ids=ids, params=config_params(cfg, SPP_QUICK))
        quick = acc.copy()
        # continue the same accumulate buffer to the full sample count (running mean, timestamps continue)
        rest = cfg["spp"] - SPP_QUICK
        acc, st_r = o.render(W, H, spp=rest, first_timestamp=SPP_QUICK, accum=acc, pixel_ids=ids,
                             params=config_params(cfg, rest, SPP_QUICK))
        out[name + "/ids"] = ids
        out[name + "/quick"] = quick
        out[name + "/full"] = acc
        out[name + "/counts_quick"] = np.array(
            [st_q["extension_rays"], st_q["shadow_rays"], st_q["shaded_vertices"], st_q["samples"]], np.uint64)
        out[name + "/counts_full"] = np.array(
            [st_q[k] + st_r[k] for k in ("extension_rays", "shadow_rays", "shaded_vertices", "samples")], np.uint64)
        out[name + "/triangles"] = np.array([st_q["num_triangles"]], np.uint64)
        print("%-10s %7d tris  %dx%d  %d px  spp %d+%d  %.1f s  rays/sample %.2f" % (
            name, st_q["num_triangles"], W, H, len(ids), SPP_QUICK, rest, time.time() - t0,
            (out[name + "/counts_full"][0] + out[name + "/counts_full"][1]) / out[name + "/counts_full"][3]), flush=True)
        o.close()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
