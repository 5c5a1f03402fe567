#!/usr/bin/env python3
"""The statistics of statistical_outlier_removal alone (pcrs_sor_select_dev: threshold from the exact
left-to-right f32 sums, then the mask) on vectors of means of several lengths."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
import torch
from pointclouds_rs.dist import HipEngine
eng = HipEngine.shared(0)
for n in [int(float(a)) for a in sys.argv[1:]] or [1_000_000, 8_000_000]:
    rng = np.random.default_rng(3)
    mean = (0.8 + 0.4 * rng.random(n)).astype(np.float32)
    d = eng.from_numpy(mean)
    keep = eng.empty(n, "uint8")
    ts = []
    for rep in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        thr, ok, n_keep = eng.select(d, 1.0, d, keep)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f"n = {n}: threshold + mask {1e3 * np.median(ts[2:]):.3f} ms (thr {thr:.6f}, kept {n_keep})", flush=True)
