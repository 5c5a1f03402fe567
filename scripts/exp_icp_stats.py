#!/usr/bin/env python3
"""Experiment: how many 1-NN queries of the ICP benchmark's first iteration each search stage
answers (cell kernel, wave retries, brute force)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
from pointclouds_rs import hip, synth

n = 100_000
tgt = synth.hemisphere(n, 99, 5.0)
P = np.stack(tgt, 1)
for ang, tr in ((0.05, (0.3, -0.2, 0.1)), (0.0, (0.0, 0.0, 0.0)), (0.002, (0.01, 0.0, 0.0))):
    S = (P @ synth.rot_z(ang).T + np.array(tr, dtype=np.float32)).astype(np.float32)
    ix = hip.Index(*tgt, k_hint=1)
    res = ix.find_correspondences(np.ascontiguousarray(S[:, 0]), np.ascontiguousarray(S[:, 1]),
                                        np.ascontiguousarray(S[:, 2]))
    print(ang, tr, "stages", ix.last_stats(), "pairs", len(res[0]), flush=True)
    ix.close()
