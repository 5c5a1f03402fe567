#!/usr/bin/env python3
"""Experiment: pcrs_sor_select_dev over n means (default 50 M) for the section length of the long
statistics path given in PCRS_STATS_SECTION (chunks of 256 elements); prints time and threshold
(which must not depend on the section length)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np, torch
from pointclouds_rs.dist import HipEngine

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
eng = HipEngine(0)
g = torch.Generator(device="cpu").manual_seed(7)
m = (torch.rand(n, generator=g) * 0.8 + 0.9).to(eng.device)
keep = eng.empty(1_000_000, "uint8")
ts = []
for rep in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    thr, ok, nk = eng.select(m, 1.0, m[:1_000_000], keep)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"section {os.environ.get('PCRS_STATS_SECTION', 'default')} n {n}: select {np.median(ts[2:]) * 1e3:.3f} ms  "
      f"thr {np.float32(thr).view(np.uint32):#x} kept {nk}", flush=True)
