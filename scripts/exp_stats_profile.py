#!/usr/bin/env python3
"""Experiment: 20 calls of pcrs_sor_select_dev over n means (default 8 M) for a kernel trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import torch
from pointclouds_rs.dist import HipEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
eng = HipEngine(0)
g = torch.Generator(device="cpu").manual_seed(7)
m = (torch.rand(n, generator=g) * 0.8 + 0.9).to(eng.device)
keep = eng.empty(1_000_000, "uint8")
for rep in range(20):
    eng.select(m, 1.0, m[:1_000_000], keep)
torch.cuda.synchronize()
