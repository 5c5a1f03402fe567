#!/usr/bin/env python3
"""Experiment: euclidean_cluster with a thread or a wave per point (PCRS_CLUSTER_WAVE_MAX selects
the largest input that gets a wave per point) on sparse and on crowded clouds of several sizes."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    for wm in ("0", "4000000000"):
        env = dict(os.environ, PCRS_CLUSTER_WAVE_MAX=wm)
        subprocess.check_call([sys.executable, __file__, "run"], env=env)
    sys.exit(0)
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
from pointclouds_rs import hip
rng = np.random.default_rng(5)
cases = {}
cases["uniform 1M thr 0.5"] = (rng.uniform(0, 100, (1_000_000, 3)).astype(np.float32), 0.5)
cases["uniform 1M thr 1.2"] = (rng.uniform(0, 100, (1_000_000, 3)).astype(np.float32), 1.2)
c = rng.uniform(-50, 50, (40, 3))
blobs = (c[rng.integers(0, 40, 200_000)] + rng.uniform(-0.8, 0.8, (200_000, 3))).astype(np.float32)
cases["40 blobs 200K thr 0.8"] = (blobs, 0.8)
blobs2 = (c[rng.integers(0, 40, 1_000_000)] + rng.normal(0, 1.5, (1_000_000, 3))).astype(np.float32)
cases["40 gaussian blobs 1M thr 0.3"] = (blobs2, 0.3)
for nb in (5_000, 20_000, 60_000):
    cases[f"40 blobs {nb} thr 0.8"] = ((c[rng.integers(0, 40, nb)] + rng.uniform(-0.8, 0.8, (nb, 3))).astype(np.float32), 0.8)
    cases[f"uniform {nb} thr 1.2 in 30^3"] = (rng.uniform(0, 30, (nb, 3)).astype(np.float32), 1.2)
for name, (p, thr) in cases.items():
    x, y, z = (np.ascontiguousarray(p[:, a]) for a in range(3))
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        cl = hip.euclidean_cluster(x, y, z, thr, 10, 10_000_000)
        best = min(best, time.perf_counter() - t0)
    print(f"wave_max={os.environ['PCRS_CLUSTER_WAVE_MAX']:>10s} {name:32s} {1e3*best:8.2f} ms, {len(cl)} clusters, "
          f"largest {len(cl[0]) if cl else 0}", flush=True)
