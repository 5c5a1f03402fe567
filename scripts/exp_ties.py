#!/usr/bin/env python3
"""Experiment: clouds whose coordinates are quantised (sensor resolution, voxel centres), where
equal and nearly equal neighbour distances are common.  Prints the time of SOR + normals, the
split of the search over its stages and a checksum of the results, so that two builds or two
kernels (PCRS_ONE_WALK=0 / 1) can be set side by side: the checksums must agree."""
import os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
from pointclouds_rs import hip, synth

def run(name, x, y, z):
    x, y, z = (np.ascontiguousarray(c, dtype=np.float32) for c in (x, y, z))
    hip.sor_normals(x, y, z, 16, 1.0, 16)
    t0 = time.perf_counter()
    keep, nk, thr, nrm = hip.sor_normals(x, y, z, 16, 1.0, 16)
    t1 = time.perf_counter()
    ix = hip.Index(x, y, z, k_hint=17)
    idx, dist, cnt = ix.knn(x[:200000], y[:200000], z[:200000], 16)
    st = ix.last_stats(); ix.close()
    crc = zlib.crc32(keep.tobytes())
    for a in nrm: crc = zlib.crc32(np.ascontiguousarray(a).tobytes(), crc)
    crc = zlib.crc32(idx.tobytes(), crc); crc = zlib.crc32(dist.tobytes(), crc)
    print(f"{name}: n={len(x)} sor+normals {1e3*(t1-t0):.2f} ms kept {nk} thr {thr:.6f} knn stages {st} crc {crc:08x}", flush=True)

n = 1_000_000
x, y, z = synth.uniform_cloud(n, 7, 0.0, 100.0)
for q in (0.01, 0.05, 0.25):
    run(f"uniform, coordinates rounded to {q}", np.round(x / q) * q, np.round(y / q) * q, np.round(z / q) * q)
rng = np.random.default_rng(5)
px, py = rng.uniform(0, 300, n), rng.uniform(0, 300, n)
pz = 0.02 * rng.normal(size=n)
run("plane with 2 cm noise, rounded to 1 cm", np.round(px, 2), np.round(py, 2), np.round(pz, 2))
g = np.stack(np.meshgrid(np.arange(100), np.arange(100), np.arange(100), indexing="ij"), -1).reshape(-1, 3)
run("integer lattice 100^3", g[:, 0], g[:, 1], g[:, 2])
