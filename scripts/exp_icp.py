#!/usr/bin/env python3
"""Experiment: the ICP benchmark loop alone (for kernel traces and counter passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd")); sys.path.insert(0, ROOT)
import bench
from pointclouds_rs import synth
from pointclouds_rs.dist import HipEngine
eng = HipEngine(0)
r = bench.bench_icp(eng, synth, int(sys.argv[1]) if len(sys.argv) > 1 else 50)
print(r)
import numpy as np
