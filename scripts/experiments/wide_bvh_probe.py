#!/usr/bin/env python3
"""Dumps the world-space triangles of a synthetic scene for wide_bvh_probe.cpp (CPU experiment, not product code).
usage: wide_bvh_probe.py interior|caustics|materials <target_tris> out.bin"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from gpuspectral_amd import scenes

kind, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
sc = {"interior": lambda: scenes.interior(n, seed=7), "caustics": lambda: scenes.caustics(n, seed=11),
      "materials": lambda: scenes.cornell_materials()}[kind]()
tris = []
for inst in sc.instances:
    m = inst["transform"].reshape(4, 4).astype(np.float64)  # m[c] = column c
    p = sc.positions[inst["first_vertex"]: inst["first_vertex"] + inst["vertex_count"]].astype(np.float64)
    w = p[:, 0:1] * m[0, :3] + p[:, 1:2] * m[1, :3] + p[:, 2:3] * m[2, :3] + m[3, :3]
    tris.append(w.astype(np.float32))
t = np.concatenate(tris).reshape(-1, 9)
t.tofile(out)
print(len(t), "triangles ->", out)
