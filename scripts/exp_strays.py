#!/usr/bin/env python3
"""Experiment: queries that leave the 27-cell cube by far.  (a) 1-NN correspondences of a 100 K
scan that overlaps a 5 M-point target only in part (the rest of it lies up to 30 units outside);
(b) SOR k = 16 on a 2 M-point surface with 20 K strays above it.  PCRS_HIP_LIB selects the
library, so two builds can be compared."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
from pointclouds_rs import hip, synth

rng = np.random.default_rng(3)
tx, ty, tz = synth.uniform_cloud(5_000_000, 31, 0, 100)
src = rng.uniform(0, 100, (100_000, 3)).astype(np.float32)
src[:, 0] += 30.0
ix = hip.Index(tx, ty, tz, k_hint=1)
for rep in range(3):
    t0 = time.perf_counter()
    t, d, n = ix.find_correspondences(src[:, 0], src[:, 1], src[:, 2], np.inf)
    t1 = time.perf_counter()
print(f"lib {os.environ.get('PCRS_HIP_LIB', 'default')}: correspondences 100 K x 5 M, 30 % of the scan outside: "
      f"{1e3*(t1-t0):.2f} ms, stages {ix.last_stats()}, checksum {int(t.astype(np.uint64).sum())}", flush=True)
ix.close()
n_s, n_o = 2_000_000, 20_000
surf = np.stack([rng.uniform(-100, 100, n_s), rng.uniform(-100, 100, n_s), rng.uniform(-0.05, 0.05, n_s)], 1)
stray = np.stack([rng.uniform(-100, 100, n_o), rng.uniform(-100, 100, n_o), rng.uniform(1, 30, n_o)], 1)
p = np.concatenate([surf, stray]).astype(np.float32)
x, y, z = (np.ascontiguousarray(p[:, a]) for a in range(3))
for rep in range(3):
    t0 = time.perf_counter()
    keep, n_keep, thr, mean = hip.sor(x, y, z, 16, 1.0)
    t1 = time.perf_counter()
print(f"   SOR k=16 on a 2 M-point surface + 20 K strays: {1e3*(t1-t0):.2f} ms, kept {n_keep}, "
      f"threshold {thr:.6f}", flush=True)
