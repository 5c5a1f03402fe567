#!/usr/bin/env python3
"""Experiment: what one rank of an N-GPU run computes per step (without the collectives): index
over all N x 1 M points, search of its own 1 M-point range, statistics over all N x 1 M means."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np, torch
from pointclouds_rs import synth
from pointclouds_rs.dist import HipEngine

eng = HipEngine(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "index"  # "index", "cells" or "slabs"
worlds = [int(sys.argv[2])] if len(sys.argv) > 2 else [1, 2, 4, 8]
for world in worlds:
    n = world * 1_000_000
    side = np.float32(100.0 * world ** (1.0 / 3.0))
    cols = [eng.from_numpy((synth.uniform01(42, n, a) * side).astype(np.float32)) for a in range(3)]
    nq = 1_000_000
    mean = eng.empty(nq); nx = eng.empty(nq); ny = eng.empty(nq); nz = eng.empty(nq); keep = eng.empty(nq, "uint8")
    mean_full = eng.empty(n); mean_full.fill_(1.0)
    point = eng.empty(nq, "int32")
    gathered = eng.empty(n); gathered.fill_(1.0)
    point_full = torch.randperm(n, device=eng.device).to(torch.int32)
    ts = []
    for rep in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ix = eng.build_index(*cols, 17)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if mode == "slabs":
            # the slab of rank `world // 2` (halo on both sides) and its index instead of `ix`
            eng.destroy_index(ix)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ix, c, axis = eng.slab_index(*cols, 17, world, world // 2)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            if c > mean.numel():
                mean = eng.empty(c); nx = eng.empty(c); ny = eng.empty(c); nz = eng.empty(c)
                keep = eng.empty(c, "uint8"); point = eng.empty(c, "int32")
            esc = eng.sor_normals_selected(ix, 16, 16, (0.0, 0.0, 0.0), point[:c], mean[:c], nx[:c],
                                           ny[:c], nz[:c])
            torch.cuda.synchronize(); t2 = time.perf_counter()
            # every rank's (point, mean) pairs arrive by all-gather; dropping them into index order
            # is this rank's part (the pairs of the other ranks are made up)
            eng.scatter(gathered, point_full, mean_full)
        elif mode == "index":
            eng.sor_normals_range(ix, 0, nq, 16, 16, (0.0, 0.0, 0.0), mean, nx, ny, nz)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            mean_full[:nq] = mean
        else:  # the run of cells of rank 0; the means of the other ranks are made up
            bounds = eng.slot_bounds(ix, world)
            c = bounds[1] - bounds[0]
            if c > mean.numel():
                mean = eng.empty(c); nx = eng.empty(c); ny = eng.empty(c); nz = eng.empty(c)
                keep = eng.empty(c, "uint8"); point = eng.empty(c, "int32")
            eng.sor_normals_slots(ix, bounds[0], c, 16, 16, (0.0, 0.0, 0.0), point[:c], mean[:c],
                                  nx[:c], ny[:c], nz[:c])
            torch.cuda.synchronize(); t2 = time.perf_counter()
            mean_full.zero_()  # the sum over ranks of these vectors is the collective's part
            eng.scatter(mean[:c], point[:c], mean_full)
            mean_full += gathered  # stands in for the other ranks' means
        thr, ok, nk = eng.select(mean_full, 1.0, mean[:c] if mode == "slabs" else mean, keep[:c] if mode == "slabs" else keep)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        eng.destroy_index(ix)
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    b, s, st = np.median(np.array(ts[1:]), 0) * 1e3
    if mode == "slabs":
        print(f"   slab of rank {world // 2}: {c} queries, {esc} escalated, axis {axis}", flush=True)
    print(f"{mode} world {world}: build {b:.3f} ms  search(1M of {n}) {s:.3f} ms  stats+mask {st:.3f} ms  total {b+s+st:.3f} ms", flush=True)
