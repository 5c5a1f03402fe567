#!/usr/bin/env python3
"""Experiment: statistics + mask of pcrs_sor_select_dev over n means, for means around 1.3 (the
f32 running sum passes 2^23, where every add rounds to a whole ulp) and around 0.013 (it does not)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np, torch
from pointclouds_rs.dist import HipEngine

eng = HipEngine(0)
g = torch.Generator(device="cpu").manual_seed(7)
base = torch.rand(8_000_000, generator=g) * 0.8 + 0.9
keep = eng.empty(1_000_000, "uint8")
for scale in (1.0, 0.01):
    for n in (1_000_000, 2_000_000, 4_000_000, 8_000_000):
        m = (base[:n] * scale).to(eng.device)
        ts = []
        for rep in range(12):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            thr, ok, nk = eng.select(m, 1.0, m[:1_000_000], keep)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(f"scale {scale} n {n}: select {np.median(ts[4:]) * 1e3:.3f} ms  thr {thr:.6g}", flush=True)
