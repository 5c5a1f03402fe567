#!/usr/bin/env python3
"""Experiment: the aerial scene at about 100 M points (scale 41.5), resident columns, split into index
build and search + statistics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
import torch
from pointclouds_rs import synth
from pointclouds_rs.dist import HipEngine
eng = HipEngine.shared(0)
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 41.5
cols = synth.aerial_scene(scale)
n = len(cols[0])
shard = [eng.from_numpy(c) for c in cols]
del cols
mean, keep = eng.empty(n), eng.empty(n, "uint8")
nx, ny, nz = eng.empty(n), eng.empty(n), eng.empty(n)
tb, ts, td = [], [], []
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix = eng.build_index(*shard, 17)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    eng.sor_normals_range(ix, 0, n, 16, 16, (250.0, 250.0, 1000.0), mean, nx, ny, nz)
    thr, ok, n_keep = eng.select(mean, 1.0, mean, keep)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    st = eng.index_stats(ix)
    t3 = time.perf_counter()
    eng.destroy_index(ix)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    tb.append(t1 - t0); ts.append(t2 - t1); td.append(t4 - t3)
print(f"{n} points: build {1e3 * np.median(tb[1:]):.1f} ms, search + statistics {1e3 * np.median(ts[1:]):.1f} ms, "
      f"release {1e3 * np.median(td[1:]):.1f} ms, stages {st}, kept {n_keep}", flush=True)
