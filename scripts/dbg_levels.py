#!/usr/bin/env python3
"""Debug aid: SOR means of a ring scan with the finer levels on, against the oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_ffi
from pointclouds_rs import hip, synth
o = oracle_ffi.load()
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x, y, z = synth.ring_scan(sweeps)
t = time.time(); rkeep, rn, rthr, rmean = o.sor(x, y, z, 16, 1.0, algo=1); print("oracle", time.time() - t, rn, rthr, flush=True)
keep, nk, thr, mean = hip.sor(x, y, z, 16, 1.0)
bad = np.flatnonzero(mean.view(np.uint32) != rmean.view(np.uint32))
print("hip", nk, thr, "bad means", len(bad))
for i in bad[:10]:
    print(i, x[i], y[i], z[i], mean[i], rmean[i])
keep2, nk2, thr2, nrm = hip.sor_normals(x, y, z, 16, 1.0, 16, (0.0, 0.0, 1.73))
print("fused", nk2, thr2)
