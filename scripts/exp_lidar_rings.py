#!/usr/bin/env python3
"""Experiment: a spinning-LiDAR shaped scan (64 beams x 2048 azimuths on a ground plane, 1.73 m
sensor height: dense rings nearby, sparse ones far out, density ~ 1/r^2) through SOR k = 16 and
normals k = 16, with the split of the search over its stages.  A uniform grid sized for the mean
density cannot certify the sparse rings in 27 cells; they take the retry stage."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
from pointclouds_rs import hip

rng = np.random.default_rng(11)
for sweeps in ([int(a) for a in sys.argv[1:]] or [1, 8]):
    beams, az = 64, 2048
    elev = np.deg2rad(np.linspace(-24.8, -1.0, beams))      # below the horizon: ground hits
    r = 1.73 / np.tan(-elev)                                # 3.7 m ... 99 m
    parts = []
    for s in range(sweeps):                                 # several sweeps of a moving sensor
        a = np.linspace(0, 2 * np.pi, az, endpoint=False) + rng.uniform(0, 2 * np.pi / az)
        rr = (r[:, None] * (1 + rng.normal(0, 2e-3, (beams, az)))).ravel()
        aa = np.broadcast_to(a, (beams, az)).ravel()
        parts.append(np.stack([rr * np.cos(aa) + 0.8 * s, rr * np.sin(aa), rng.normal(0, 0.02, rr.size)], 1))
    p = np.concatenate(parts).astype(np.float32)
    x, y, z = (np.ascontiguousarray(p[:, i]) for i in range(3))
    ix = hip.Index(x, y, z, k_hint=17)
    for rep in range(3):
        t0 = time.perf_counter()
        idx, d, c = ix.knn(x, y, z, 17)
        t1 = time.perf_counter()
    st = ix.last_stats()
    ix.close()
    for rep in range(3):
        t2 = time.perf_counter()
        keep, n_keep, thr, nrm = hip.sor_normals(x, y, z, 16, 1.0, 16, (0.0, 0.0, 1.73))
        t3 = time.perf_counter()
    print(f"{sweeps} sweep(s), {len(x)} points: stages {st}; knn lists {1e3*(t1-t0):.2f} ms, "
          f"SOR + normals (host buffers) {1e3*(t3-t2):.2f} ms = {len(x)/(t3-t2)/1e6:.0f} Mpoints/s, kept {n_keep}", flush=True)
