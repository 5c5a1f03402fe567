#!/usr/bin/env python3
"""Experiment: bench.py's aerial line at a chosen size (points), on a fresh engine."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import bench
from pointclouds_rs.dist import HipEngine, Collectives
eng = HipEngine.shared(0)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 24_100_000
print(bench.bench_aerial(eng, Collectives(), n, types.SimpleNamespace(steps=3)), flush=True)
