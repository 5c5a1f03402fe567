#!/usr/bin/env python3
"""Experiment: SOR k = 16 + normals k = 16 on the aerial scene (synth.aerial_scene: terrain surface,
buildings, trees) at several scales, resident columns, with the split of the search over its stages."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
import torch
from pointclouds_rs import synth
from pointclouds_rs.dist import Collectives, HipEngine, sor_normals_sharded
eng = HipEngine.shared(0)
coll = Collectives(None)
for scale in [float(a) for a in sys.argv[1:]] or [1.0, 4.0]:
    cols = synth.aerial_scene(scale)
    n = len(cols[0])
    shard = [eng.from_numpy(c) for c in cols]
    stats = sor_normals_sharded(eng, coll, shard, n, 16, 1.0, 16, (250.0, 250.0, 1000.0), want_stats=True)["stats"]
    ts = []
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = sor_normals_sharded(eng, coll, shard, n, 16, 1.0, 16, (250.0, 250.0, 1000.0))
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ix = eng.build_index(*shard, 17)
    lv = (eng.lib.pcrs_index_levels, )
    import ctypes as C
    a, b = C.c_uint32(0), C.c_uint32(0)
    eng.lib.pcrs_index_levels(ix, C.byref(a), C.byref(b))
    eng.destroy_index(ix)
    print(f"scale {scale}: {n} points, {1e3 * np.median(ts[1:]):.2f} ms = {n / np.median(ts[1:]) / 1e6:.0f} Mpoints/s; "
          f"stages {stats}; levels finer {a.value} coarser {b.value}", flush=True)
