#!/usr/bin/env python3
"""Experiment: a spinning-LiDAR shaped cloud (64 beams x 1800 azimuths from a sensor 1.73 m above
flat ground, ranges cut at 80 m, plus two walls): the point density falls with the square of the
range, which a uniform grid does not like.  SOR k=20 and normals k=16 through the host API, and
which search stage answered how many queries."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
from pointclouds_rs import hip

rng = np.random.default_rng(0)
elev = np.deg2rad(np.linspace(-24.8, 2.0, 64))
az = np.linspace(0.0, 2 * np.pi, 1800, endpoint=False)
E, A = np.meshgrid(elev, az, indexing="ij")
h = 1.73
with np.errstate(divide="ignore"):
    r_ground = np.where(E < 0, h / np.tan(-E), np.inf)
r = np.minimum(r_ground, 80.0)
# two walls: x = 15 (|y| < 20) and y = -10 (0 < x < 30)
cosA, sinA = np.cos(A), np.sin(A)
with np.errstate(divide="ignore", invalid="ignore"):
    rw1 = np.where(cosA > 0, 15.0 / (cosA * np.cos(E)), np.inf)
    hit1 = np.abs(rw1 * np.cos(E) * sinA) < 20
    rw2 = np.where(sinA < 0, -10.0 / (sinA * np.cos(E)), np.inf)
    x2 = rw2 * np.cos(E) * cosA
    hit2 = (x2 > 0) & (x2 < 30)
r = np.minimum(r, np.where(hit1, rw1, np.inf))
r = np.minimum(r, np.where(hit2, rw2, np.inf))
ok = np.isfinite(r) & (r < 80.0)
r = r + rng.normal(0, 0.02, r.shape)
x = (r * np.cos(E) * cosA)[ok].astype(np.float32)
y = (r * np.cos(E) * sinA)[ok].astype(np.float32)
z = (h + r * np.sin(E))[ok].astype(np.float32)
print("points", len(x), "range", float(np.hypot(x, y).min()), float(np.hypot(x, y).max()))
hip.sor(x, y, z, 20, 2.0)
for rep in range(3):
    t0 = time.perf_counter(); keep, nk, thr, _ = hip.sor(x, y, z, 20, 2.0)
    t1 = time.perf_counter(); nrm = hip.normals(x, y, z, 16)
    t2 = time.perf_counter()
    print(f"sor k=20 {1e3*(t1-t0):.2f} ms  normals k=16 {1e3*(t2-t1):.2f} ms  kept {nk}", flush=True)
ix = hip.Index(x, y, z, k_hint=21)
idx, dist, cnt = ix.knn(x, y, z, 21)
print("stages", ix.last_stats())
ix.close()
