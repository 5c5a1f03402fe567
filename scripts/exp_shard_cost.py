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
c4 = len(sys.argv) > 3 and sys.argv[3] == "c4"  # one 100 M-point tile divided among the ranks
for world in worlds:
    if c4:
        n = 100_000_000
        scale = (500.0, 500.0, 20.0)
        cols = [eng.from_numpy((synth.uniform01(9, n, a) * np.float32(scale[a])).astype(np.float32)) for a in range(3)]
        nq = n // world + n // 50
    else:
        n = world * 1_000_000
        side = np.float32(100.0 * world ** (1.0 / 3.0))
        cols = [eng.from_numpy((synth.uniform01(42, n, a) * side).astype(np.float32)) for a in range(3)]
        nq = 1_000_000
    mean = eng.empty(nq); nx = eng.empty(nq); ny = eng.empty(nq); nz = eng.empty(nq); keep = eng.empty(nq, "uint8")
    mean_full = eng.empty(n); mean_full.fill_(1.0)
    point = eng.empty(nq, "int32")
    gathered = eng.empty(n); gathered.fill_(1.0)
    recv_point = torch.randperm(n // world, device=eng.device).to(torch.int32)
    ts = []
    for rep in range(4 if c4 else 6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode != "slabs":
            ix = eng.build_index(*cols, 17)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if mode == "slabs":
            # the slab of rank `world // 2` (halo on both sides) and its index
            ix, c, axis = eng.slab_index(*cols, 17, world, world // 2)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            if c > mean.numel():
                mean = eng.empty(c); nx = eng.empty(c); ny = eng.empty(c); nz = eng.empty(c)
                keep = eng.empty(c, "uint8"); point = eng.empty(c, "int32")
            esc = eng.sor_normals_selected(ix, 16, 16, (0.0, 0.0, 0.0), point[:c], mean[:c], nx[:c],
                                           ny[:c], nz[:c])
            torch.cuda.synchronize(); t2 = time.perf_counter()
            # this rank's part of the exchange of the means: its (point, mean) pairs bucketed by the
            # owner of the index range (all-to-all), the pairs it receives dropped into its own
            # range (made up here: as many as it owns), the ranges all-gathered into mean_full
            send_point = eng.empty(c, "int32"); send_mean = eng.empty(c)
            eng.route_pairs(point[:c], mean[:c], n, world, send_point, send_mean)
            own = n // world
            eng.scatter_range(gathered[:own], recv_point[:own], 0, mean_full[:own])
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
    print(f"{mode} world {world}: build {b:.3f} ms  search({nq if mode == 'index' else c} of {n}) {s:.3f} ms  stats+mask {st:.3f} ms  total {b+s+st:.3f} ms", flush=True)
