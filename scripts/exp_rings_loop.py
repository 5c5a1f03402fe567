#!/usr/bin/env python3
"""A ring scan (synth.ring_scan) or blob cloud through SOR + normals a few times (for kernel traces):
host buffers through pcrs_sor_normals, then resident columns through the calls of a bench step."""
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
line = f"{kind}: {len(x)} points, SOR + normals (host buffers) median {1e3 * np.median(ts[1:]):.2f} ms, kept {n_keep}"
if "dev" in sys.argv[2:]:
    import torch
    from pointclouds_rs.dist import Collectives, HipEngine, sor_normals_sharded
    eng = HipEngine.shared(0)
    shard = [eng.from_numpy(c) for c in (x, y, z)]
    coll = Collectives(None)
    ts = []
    for rep in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = sor_normals_sharded(eng, coll, shard, len(x), 16, 1.0, 16, (0.0, 0.0, 1.73))
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    assert r["n_keep"] == n_keep
    line += f"; resident columns median {1e3 * np.median(ts[2:]):.2f} ms (min {1e3 * min(ts[2:]):.2f})"
print(line)
