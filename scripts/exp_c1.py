#!/usr/bin/env python3
"""Experiment: the reference's own benchmark case alone (C1: SOR k=10 std_mul=1.0 on 100 K uniform
points in [0, 100)^3, host buffers in and out), for kernel or API traces of that one call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
from pointclouds_rs import hip, synth
x, y, z = synth.uniform_cloud(100_000, 42, 0.0, 100.0)
for _ in range(3): hip.sor(x, y, z, 10, 1.0)
ts = []
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
    t0 = time.perf_counter(); keep, nk, thr, _ = hip.sor(x, y, z, 10, 1.0); ts.append(time.perf_counter() - t0)
print(f"C1: SOR k=10 on 100 K points, host buffers: median {1e3*np.median(ts):.3f} ms, min {1e3*min(ts):.3f} ms, kept {nk}")
