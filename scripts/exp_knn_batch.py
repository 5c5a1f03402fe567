#!/usr/bin/env python3
"""Experiment: pcrs_knn_batch_dev (index + distance lists) on a resident 1 M-point cloud."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np, torch
from pointclouds_rs import synth
from pointclouds_rs.dist import HipEngine
eng = HipEngine(0)
n = 1_000_000
cols = [eng.from_numpy(c) for c in synth.uniform_cloud(n, 42)]
for k in (1, 10, 16, 32):
    ix = eng.build_index(*cols, k)
    idx = eng.empty(n * k, "int32"); dist = eng.empty(n * k); cnt = eng.empty(n, "int32")
    ts = []
    for rep in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.ctx.check(eng.lib.pcrs_knn_batch_dev(ix, *[eng._p(c) for c in cols], n, k, eng._p(idx), eng._p(dist), eng._p(cnt)))
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    eng.destroy_index(ix)
    print(f"k={k}: {1e3*np.median(ts[1:]):.3f} ms  ({n/np.median(ts[1:])/1e6:.0f} Mqueries/s)", flush=True)
