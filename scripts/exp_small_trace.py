#!/usr/bin/env python3
"""Experiment: 200 calls of SOR k=10 on 100 K uniform points through the host API (for traces)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
from pointclouds_rs import hip, synth
x, y, z = synth.uniform_cloud(100_000, 42, 0.0, 100.0)
hip.sor(x, y, z, 10, 1.0)
t0 = time.perf_counter()
for _ in range(200):
    hip.sor(x, y, z, 10, 1.0)
print("ms per call", 1e3 * (time.perf_counter() - t0) / 200)
