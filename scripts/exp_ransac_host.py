#!/usr/bin/env python3
"""Experiment: the host-buffer plane RANSAC call on a frame-sized cloud (45 K points, 500
triples), wall time per call; under `rocprofv3 --hip-trace --stats` it shows which runtime calls
the time of such a small call goes to."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
import pointclouds_rs as pc
from pointclouds_rs import hip

rng = np.random.default_rng(3)
n = 45_000
x = rng.uniform(-40, 40, n).astype(np.float32); y = rng.uniform(-40, 40, n).astype(np.float32)
z = (0.02 * rng.normal(size=n)).astype(np.float32)
smp = pc.sample_triples(n, 500, 1234)
for _ in range(3): hip.ransac_plane_samples(x, y, z, smp, 0.15)
ts = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
    t0 = time.perf_counter(); r = hip.ransac_plane_samples(x, y, z, smp, 0.15); ts.append(time.perf_counter() - t0)
print(f"ransac_plane_samples, {n} points x {len(smp)} triples: median {1e3*np.median(ts):.3f} ms, min {1e3*min(ts):.3f} ms, inliers {int(r[2].sum()) if hasattr(r[2], 'sum') else r[2]}")
