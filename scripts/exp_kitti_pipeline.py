#!/usr/bin/env python3
"""Experiment: the obstacle-detection pipeline of examples/python/kitti_obstacle_detection.py
(voxel 0.15 -> SOR k=20 x2.0 -> RANSAC ground 0.15 / 500 -> drop ground -> clusters 0.8 / 10 /
20000) on eight synthetic frames, one after the other on one GPU, through the host-buffer API
(NumPy in, NumPy out at every stage, as the reference's Python example works)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
import pointclouds_rs as pc
from pointclouds_rs import hip, synth

frames = [synth.kitti_scene(seed) for seed in range(42, 50)]
names = ("voxel", "sor", "ransac", "cluster")
for rep in range(3):
    tot = np.zeros(4)
    t_all = time.perf_counter()
    for x, y, z in frames:
        t0 = time.perf_counter()
        g = hip.voxel_downsample(x, y, z, 0.15)
        t1 = time.perf_counter()
        keep, n_keep, thr, _ = hip.sor(*g, 20, 2.0)
        c = [col[keep] for col in g]
        t2 = time.perf_counter()
        smp = pc.sample_triples(len(c[0]), 500, 1234)
        nrm, d, inl, best = hip.ransac_plane_samples(*c, smp, 0.15)
        mask = np.ones(len(c[0]), dtype=bool)
        mask[inl] = False
        obst = [col[mask] for col in c]
        t3 = time.perf_counter()
        cl = hip.euclidean_cluster(*obst, 0.8, 10, 20000)
        t4 = time.perf_counter()
        tot += (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
    t_all = time.perf_counter() - t_all
    print(f"rep {rep}: {len(frames)} frames of {len(frames[0][0])} points in {1e3*t_all:.2f} ms "
          f"({len(frames)/t_all:.0f} frames/s); per frame: " +
          ", ".join(f"{n} {1e3*t/len(frames):.3f} ms" for n, t in zip(names, tot)) +
          f"; last frame: {len(g[0])} voxels, {n_keep} kept, {len(inl)} ground, {len(cl)} clusters", flush=True)
