#!/usr/bin/env python3
"""Experiment: the obstacle pipeline over many frames in a row, time per block of 100 frames (does
anything grow with the number of calls?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
from pointclouds_rs import hip, synth
x, y, z = synth.kitti_scene(42)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
ts = []
for b in range(n // 100):
    t0 = time.perf_counter()
    for _ in range(100):
        out = hip.obstacle_pipeline(x, y, z, 0.15, 20, 2.0, 0.15, 500, 1234, 0.8, 10, 20000, want_obstacles=False, device=0)
    ts.append((time.perf_counter() - t0) * 10.0)
print("ms per frame, per block of 100 frames:", " ".join(f"{t:.3f}" for t in ts), flush=True)
