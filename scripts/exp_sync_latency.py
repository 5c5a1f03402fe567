#!/usr/bin/env python3
"""Experiment: wall time of one small device-to-host read-back + stream synchronisation on an
idle stream (pcrs_index_last_stats), against torch's own .item() round trip."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np, torch
from pointclouds_rs import synth
from pointclouds_rs.dist import HipEngine

eng = HipEngine(0)
n = 100_000
cols = [eng.from_numpy((synth.uniform01(1, n, a) * 50).astype(np.float32)) for a in range(3)]
ix = eng.build_index(*cols, 17)
mean = eng.empty(n)
eng.sor_normals_range(ix, 0, n, 16, 0, (0, 0, 0), mean, None, None, None)
torch.cuda.synchronize()
ts = []
for _ in range(200):
    t0 = time.perf_counter(); eng.index_stats(ix); ts.append(time.perf_counter() - t0)
print("library read-back + sync: median %.1f us" % (1e6 * np.median(ts)))
t = torch.zeros(1, device=eng.device)
ts = []
for _ in range(200):
    t0 = time.perf_counter(); t.item(); ts.append(time.perf_counter() - t0)
print("torch .item(): median %.1f us" % (1e6 * np.median(ts)))
ts = []
for _ in range(200):
    t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("idle synchronize: median %.1f us" % (1e6 * np.median(ts)))
ts = []
for _ in range(50):
    torch.cuda.synchronize(); t0 = time.perf_counter(); b = eng.build_index(*cols, 17); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0); eng.destroy_index(b)
print("build 100K: median %.1f us" % (1e6 * np.median(ts)))
