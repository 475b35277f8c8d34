#!/usr/bin/env python3
"""Regenerates the committed golden fixtures under tests/golden/.

The reference has no tests, golden vectors or runnable build, so
these fixtures are produced by this repository's CPU oracle and frozen here as
regression pins; the fixtures that come from the reference tree itself (third-party
ground-truth renders the author shipped) are made by tests/golden/make_tungsten.py.

  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from gpuspectral_amd import abi, scenes  # noqa: E402
from oracle import mitsuba_loader as ml  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def bsdf_cases(sc, n_wo=24, seed=123):
    """(handle, wo, seed) triples covering every BSDF record of the scene."""
    rng = np.random.RandomState(seed)
    cases = []
    for t, arr in enumerate(sc.bsdfs):
        for i in range(len(arr)):
            for _ in range(n_wo):
                wo = rng.normal(size=3)
                wo /= np.linalg.norm(wo)
                if t != abi.BSDF_SMOOTH_DIELECTRIC:
                    wo[2] = abs(wo[2])
                cases.append((abi.bsdf_handle(t, i), wo.astype(np.float32), int(rng.randint(0, 2**31 - 1))))
    return cases


def main():
    cornell = ml.load_scene(os.path.join(HERE, "cornell-box", "scene.xml"))
    o = orc.Oracle(cornell)
    # BASELINE config 1: Cornell 128x128, 1 spp, timestamp 0
    img, st = o.render(128, 128, spp=1)
    np.save(os.path.join(HERE, "cornell_128_1spp.npy"), img[:, :3].reshape(128, 128, 3).astype(np.float32))
    # first-hit map of the primary rays
    rays = np.zeros((128 * 128, 8), np.float32)
    for y in range(128):
        for x in range(128):
            r = o.primary_ray(128, 128, x, y)
            rays[y * 128 + x, 0:3] = r[:3]
            rays[y * 128 + x, 4:7] = r[3:]
    rays[:, 7] = 1e10
    hits = o.trace(rays)
    np.savez_compressed(os.path.join(HERE, "cornell_first_hit_128.npz"), t=hits["t"], prim=hits["prim"])
    # per-BSDF sample / eval tables on the full-material scene
    mats = scenes.cornell_materials(8)
    om = orc.Oracle(mats)
    cases = bsdf_cases(mats)
    handles = np.array([c[0] for c in cases], np.uint32)
    wos = np.stack([c[1] for c in cases])
    seeds = np.array([c[2] for c in cases], np.uint32)
    samples = np.stack([om.bsdf_sample(int(h), w, int(s)) for h, w, s in cases])
    wis = samples[:, 0:3].copy()
    evals = np.stack([om.bsdf_eval(int(h), w, wi) for (h, w, _), wi in zip(cases, wis)])
    lights = np.stack([om.sample_light(w * 0.5 + np.array([0, 1, 0], np.float32), int(s)) for _, w, s in cases[:64]])
    np.savez_compressed(os.path.join(HERE, "bsdf_vectors.npz"), handles=handles, wo=wos, seeds=seeds,
                        samples=samples.view(np.uint32), evals=evals.view(np.uint32), lights=lights.view(np.uint32))
    # full-material render (config 2 at test size)
    img2, _ = om.render(64, 64, spp=4)
    np.save(os.path.join(HERE, "materials_64_4spp.npy"), img2[:, :3].reshape(64, 64, 3).astype(np.float32))
    # (the reference-held Tungsten images are linearised by tests/golden/make_tungsten.py)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
