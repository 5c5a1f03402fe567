#!/usr/bin/env python3
"""Experiment: euclidean_cluster and voxel_downsample (both sort their points by cell keys) on
clouds of a few thousand to a few ten thousand points, for the limit below which one workgroup sorts
in one launch (PCRS_SORT_SMALL=<keys>)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
from pointclouds_rs import hip, synth
for n in (2000, 6000, 12000, 24000, 48000):
    side = 30.0 * (n / 10000.0) ** (1 / 3)
    x, y, z = [(synth.uniform01(5, n, a) * np.float32(side)).astype(np.float32) for a in range(3)]
    tv, tc = [], []
    for rep in range(12):
        t0 = time.perf_counter(); hip.voxel_downsample(x, y, z, 0.5); tv.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); hip.euclidean_cluster(x, y, z, 0.8, 2, 100000); tc.append(time.perf_counter() - t0)
    print(f"n = {n}: voxel_downsample {1e3 * np.median(tv[2:]):.3f} ms, euclidean_cluster {1e3 * np.median(tc[2:]):.3f} ms", flush=True)
