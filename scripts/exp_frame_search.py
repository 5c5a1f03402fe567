#!/usr/bin/env python3
"""Experiment: how the neighbour search of the SOR stage (k = 20) splits over its stages on a
voxelised LiDAR-shaped frame, and what the search costs there with the host round trip excluded
from neither (pcrs_knn_batch on the frame's own points)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
from pointclouds_rs import hip, synth

for seed in (42, 43):
    x, y, z = synth.kitti_scene(seed)
    g = hip.voxel_downsample(x, y, z, 0.15)
    idx = hip.Index(*g, k_hint=21)
    for rep in range(3):
        t0 = time.perf_counter()
        _, d, _ = idx.knn(*g, 21)
        t1 = time.perf_counter()
    st = idx.last_stats()
    r = np.sqrt(g[0] ** 2 + g[1] ** 2)
    print(f"seed {seed}: {len(g[0])} voxels, stages {st}, knn {1e3*(t1-t0):.3f} ms; "
          f"21st distance quantiles {np.quantile(d[:, 20], [0.1, 0.5, 0.9, 0.99]).round(3).tolist()}; "
          f"range quantiles {np.quantile(r, [0.1, 0.5, 0.9]).round(1).tolist()}", flush=True)
    for rep in range(3):
        t0 = time.perf_counter()
        keep, n_keep, thr, _ = hip.sor(*g, 20, 2.0)
        t1 = time.perf_counter()
    print(f"   sor {1e3*(t1-t0):.3f} ms, kept {n_keep}", flush=True)
