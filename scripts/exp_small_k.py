#!/usr/bin/env python3
"""Experiment: short neighbour lists (k = 2 ... 8) over 1 M uniform points, device-resident:
normals with k neighbours and the k-NN lists themselves."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np, torch
from pointclouds_rs import synth
from pointclouds_rs.dist import HipEngine

n = 1_000_000
eng = HipEngine(0)
cols = [eng.from_numpy(c) for c in synth.uniform_cloud(n, 42)]
nx = eng.empty(n); ny = eng.empty(n); nz = eng.empty(n)
for k in (2, 3, 4, 6, 8):
    ix = eng.build_index(*cols, k)
    for rep in range(2):
        eng.sor_normals_range(ix, 0, n, 0, k, (0, 0, 0), None, nx, ny, nz)
    eng.timing(True)
    for rep in range(5):
        eng.sor_normals_range(ix, 0, n, 0, k, (0, 0, 0), None, nx, ny, nz)
    ms, launches = eng.timing_read()
    eng.timing(False)
    print(f"normals k={k}: search kernel {ms / launches:.3f} ms, stages {eng.index_stats(ix)}", flush=True)
    eng.destroy_index(ix)
