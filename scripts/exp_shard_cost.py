#!/usr/bin/env python3
"""Experiment: what one rank of an N-GPU run computes per step (without the collectives): index
over all N x 1 M points, search of its own 1 M-point range, statistics over all N x 1 M means."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np, torch
from pointclouds_rs import synth
from pointclouds_rs.dist import HipEngine

eng = HipEngine(0)
for world in (1, 2, 4, 8):
    n = world * 1_000_000
    side = np.float32(100.0 * world ** (1.0 / 3.0))
    cols = [eng.from_numpy((synth.uniform01(42, n, a) * side).astype(np.float32)) for a in range(3)]
    nq = 1_000_000
    mean = eng.empty(nq); nx = eng.empty(nq); ny = eng.empty(nq); nz = eng.empty(nq); keep = eng.empty(nq, "uint8")
    mean_full = eng.empty(n); mean_full.fill_(1.0)
    ts = []
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ix = eng.build_index(*cols, 17)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        eng.sor_normals_range(ix, 0, nq, 16, 16, (0.0, 0.0, 0.0), mean, nx, ny, nz)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        mean_full[:nq] = mean
        thr, ok, nk = eng.select(mean_full, 1.0, mean, keep)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        eng.destroy_index(ix)
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    b, s, st = np.median(np.array(ts[1:]), 0) * 1e3
    print(f"world {world}: build {b:.3f} ms  search(1M of {n}) {s:.3f} ms  stats+mask {st:.3f} ms  total {b+s+st:.3f} ms", flush=True)
