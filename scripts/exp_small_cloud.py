#!/usr/bin/env python3
"""Experiment: the reference's own benchmark sizes (BENCHMARKS.md: 100 K uniform points in
[0, 100)^3, SOR k=10 std_mul=1.0, normals k=10) through the host-buffer API, wall time per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
from pointclouds_rs import hip, synth

for n in (10_000, 100_000, 1_000_000):
    x, y, z = synth.uniform_cloud(n, 42, 0.0, 100.0)
    hip.sor(x, y, z, 10, 1.0); hip.normals(x, y, z, 10)
    ts, tn, tf = [], [], []
    for _ in range(20):
        t0 = time.perf_counter(); keep, nk, thr, _ = hip.sor(x, y, z, 10, 1.0)
        t1 = time.perf_counter(); nrm = hip.normals(x, y, z, 10)
        t2 = time.perf_counter(); hip.sor_normals(x, y, z, 10, 1.0, 10)
        t3 = time.perf_counter()
        ts.append(t1 - t0); tn.append(t2 - t1); tf.append(t3 - t2)
    print(f"n={n}: sor k=10 {1e3*np.median(ts):.3f} ms  normals k=10 {1e3*np.median(tn):.3f} ms  "
          f"fused {1e3*np.median(tf):.3f} ms  kept {nk}", flush=True)
