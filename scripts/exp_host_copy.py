#!/usr/bin/env python3
"""Experiment: wall time of the host-pointer entry points on 1 M points (index build = upload +
build, and the whole pcrs_sor_normals call), to tune the copy stager."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
from pointclouds_rs import hip, synth
x, y, z = synth.uniform_cloud(1_000_000, 42)
def med(f, reps=7):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts[2:]))
def build():
    ix = hip.Index(x, y, z, k_hint=17, algo=2); ix.close()
print("index build (upload 12 MB + build): %.3f ms" % med(build))
print("pcrs_sor_normals: %.3f ms" % med(lambda: hip.sor_normals(x, y, z, 16, 1.0, 16)))
print("pcrs_sor: %.3f ms" % med(lambda: hip.sor(x, y, z, 16, 1.0)))
