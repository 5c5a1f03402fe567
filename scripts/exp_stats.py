import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "pointclouds-rs_amd"))
import numpy as np, torch
from pointclouds_rs.dist import HipEngine
eng = HipEngine(0)
rng = np.random.default_rng(3)
n = int(float(sys.argv[1]))
m = rng.gamma(9.0, 0.13, n).astype(np.float32)
if len(sys.argv) > 2 and sys.argv[2] == "mixed":  # three quarters of the values exactly 1.0
    m[rng.random(n) < 0.75] = 1.0
d = eng.from_numpy(m)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    thr, ok = eng.threshold(d, 1.0)
    torch.cuda.synchronize(); print("threshold", thr, ok, "ms", 1e3 * (time.perf_counter() - t0), flush=True)
