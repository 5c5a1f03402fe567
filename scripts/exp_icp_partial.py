#!/usr/bin/env python3
"""Experiment: ICP on two scans that overlap only in part (the source is the target shifted by
3 units: a third of it has no counterpart), with and without a maximum correspondence distance:
with one, queries far from the target stop at the cube that covers the distance instead of
searching the whole target."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np, torch
from pointclouds_rs import _ffi, synth
from pointclouds_rs.dist import HipEngine

eng = HipEngine(0)
n = 100_000
tgt_h = synth.hemisphere(n, 99, 5.0)
P = np.stack(tgt_h, 1)
S = (P @ synth.rot_z(0.02).T + np.array([3.0, 0.1, 0.0], dtype=np.float32)).astype(np.float32)
tgt = [eng.from_numpy(c) for c in tgt_h]
src = [eng.from_numpy(np.ascontiguousarray(S[:, a])) for a in range(3)]
nrm = [eng.empty(n) for _ in range(3)]
ix = eng.build_index(*tgt, 16)
eng.sor_normals_range(ix, 0, n, 0, 16, (0.0, 0.0, 0.0), None, *nrm)
eng.destroy_index(ix)
for max_d in (float("inf"), 0.5):
    h = eng.icp_create(src, tgt, nrm)
    eng.icp_run(h, _ffi.ICP_POINT_TO_PLANE, 2, 0.0, max_d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = eng.icp_run(h, _ffi.ICP_POINT_TO_PLANE, 20, 0.0, max_d)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    eng.icp_destroy(h)
    print(f"max_d {max_d}: {1e3*dt/max(res.num_iterations,1):.3f} ms per iteration, fitness {res.fitness:.3f} rmse {res.rmse:.4f}", flush=True)
