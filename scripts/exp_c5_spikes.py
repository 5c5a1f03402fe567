import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
from pointclouds_rs import hip, synth
x, y, z = synth.kitti_scene(42)
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 600):
    t0 = time.perf_counter()
    out = hip.obstacle_pipeline(x, y, z, 0.15, 20, 2.0, 0.15, 500, 1234, 0.8, 10, 20000, want_obstacles=False, device=0)
    ts.append((time.perf_counter() - t0) * 1e3)
import numpy as np
ts = np.array(ts)
print("median %.3f ms; frames over 2 ms:" % np.median(ts[5:]), [(int(i), round(float(t), 2)) for i, t in enumerate(ts) if t > 2.0][:40])
