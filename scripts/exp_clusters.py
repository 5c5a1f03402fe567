#!/usr/bin/env python3
"""Experiment: four dense clusters in a sparse field (synth.clusters_in_field), SOR + normals on
resident columns, with the levels the index built."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
import torch
from pointclouds_rs import synth
from pointclouds_rs.dist import HipEngine
eng = HipEngine.shared(0)
cols = synth.clusters_in_field()
n = len(cols[0])
shard = [eng.from_numpy(c) for c in cols]
mean, keep = eng.empty(n), eng.empty(n, "uint8")
nx, ny, nz = eng.empty(n), eng.empty(n), eng.empty(n)
ts = []
for rep in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix = eng.build_index(*shard, 17)
    eng.sor_normals_range(ix, 0, n, 16, 16, (0.0, 0.0, 300.0), mean, nx, ny, nz)
    thr, ok, n_keep = eng.select(mean, 1.0, mean, keep)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    st = eng.index_stats(ix)
    eng.destroy_index(ix)
print(f"{n} points: {1e3 * np.median(ts[2:]):.2f} ms, stages {st}, kept {n_keep}", flush=True)
