#!/usr/bin/env python3
"""Experiment: clouds that are surfaces (a noisy plane, a sphere shell, a KITTI-like frame) through
SOR k = 16 + normals k = 16: time and the split of the search over its stages.  With
PCRS_GRID_DEBUG=1 the library prints the grid geometry of every sizing attempt."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from pointclouds_rs import hip, synth

def run(name, x, y, z):
    x, y, z = (np.ascontiguousarray(c, dtype=np.float32) for c in (x, y, z))
    hip.sor_normals(x, y, z, 16, 1.0, 16)
    t0 = time.perf_counter()
    for _ in range(5): keep, nk, thr, nrm = hip.sor_normals(x, y, z, 16, 1.0, 16)
    t1 = time.perf_counter()
    ix = hip.Index(x, y, z, k_hint=17)
    ix.knn(x, y, z, 17)
    st = ix.last_stats(); ix.close()
    print(f"{name}: n={len(x)} sor+normals {1e3*(t1-t0)/5:.2f} ms kept {nk} stages {st}", flush=True)

rng = np.random.default_rng(5)
n = 1_000_000
run("plane 300 x 300, 2 cm noise", rng.uniform(0, 300, n), rng.uniform(0, 300, n), 0.02 * rng.normal(size=n))
v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
run("sphere shell r = 50", 50 * v[:, 0], 50 * v[:, 1], 50 * v[:, 2])
run("hemisphere r = 5 (the ICP benchmark's scan), 100 K", *synth.hemisphere(100_000, 3, 5.0))
import gen
run("KITTI-like frame", *gen.kitti_scene(42))
