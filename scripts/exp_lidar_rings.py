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

from pointclouds_rs import synth

for sweeps in ([int(a) for a in sys.argv[1:]] or [1, 8]):
    x, y, z = synth.ring_scan(sweeps)
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
