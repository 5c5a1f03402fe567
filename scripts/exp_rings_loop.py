#!/usr/bin/env python3
"""A ring scan (synth.ring_scan) or blob cloud through pcrs_sor_normals a few times (for kernel traces)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
from pointclouds_rs import hip, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "rings"
x, y, z = synth.ring_scan(8) if kind == "rings" else synth.blob_in_box(100_000, 100_000)
ts = []
for rep in range(6):
    t0 = time.perf_counter()
    keep, n_keep, thr, nrm = hip.sor_normals(x, y, z, 16, 1.0, 16, (0.0, 0.0, 1.73))
    ts.append(time.perf_counter() - t0)
print(f"{kind}: {len(x)} points, SOR + normals (host buffers) median {1e3 * np.median(ts[1:]):.2f} ms, kept {n_keep}")
