#!/usr/bin/env python3
"""Experiment: neighbour-search kernel time (HIP events) for SOR-only / normals-only / fused,
optionally sweeping the grid occupancy target (PCRS_GRID_OCC is read at index build)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np, torch
from pointclouds_rs import synth
from pointclouds_rs.dist import HipEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
occs = [float(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
eng = HipEngine(0)
cols = [eng.from_numpy(c) for c in synth.uniform_cloud(n, 42)]
mean = eng.empty(n); nx = eng.empty(n); ny = eng.empty(n); nz = eng.empty(n)
for occ in occs:
    if occ > 0: os.environ["PCRS_GRID_OCC"] = str(occ)
    for name, ks, kn in (("sor16", 16, 0), ("nrm16", 0, 16), ("fused16", 16, 16), ("sor10", 10, 0), ("nrm10", 0, 10)):
        if only and name not in only: continue
        ix = eng.build_index(*cols, max(ks + 1 if ks else 0, kn))
        for rep in range(2):
            eng.sor_normals_range(ix, 0, n, ks, kn, (0, 0, 0), mean if ks else None, *( (nx, ny, nz) if kn else (None, None, None)))
        eng.timing(True)
        reps = 5
        for rep in range(reps):
            eng.sor_normals_range(ix, 0, n, ks, kn, (0, 0, 0), mean if ks else None, *( (nx, ny, nz) if kn else (None, None, None)))
        ms, launches = eng.timing_read()
        eng.timing(False)
        st = eng.index_stats(ix)
        eng.destroy_index(ix)
        print(f"occ={occ} {name}: {ms/reps:.3f} ms/query-call  ({n/(ms/reps)/1e3:.1f} Mq/s) stats={st}", flush=True)
