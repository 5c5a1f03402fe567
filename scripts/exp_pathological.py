import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "pointclouds-rs_amd"))
import numpy as np
from pointclouds_rs import hip
rng = np.random.default_rng(0)
def run(name, x, y, z):
    hip.sor(x, y, z, 16, 1.0)   # warm-up (the first call also loads the library)
    hip.normals(x, y, z, 16)
    t0 = time.perf_counter()
    keep, nk, thr, mean = hip.sor(x, y, z, 16, 1.0)
    t1 = time.perf_counter()
    nrm = hip.normals(x, y, z, 16)
    t2 = time.perf_counter()
    print(f"{name}: n={len(x)} sor {1e3*(t1-t0):.1f} ms normals {1e3*(t2-t1):.1f} ms kept {nk}", flush=True)
n = 100_000
one = np.ones(n, np.float32)
run("all-identical", one, one * 2, one * 3)
c = rng.normal(0, 1e-3, size=(n, 3)).astype(np.float32)
bg = rng.uniform(-100, 100, size=(n, 3)).astype(np.float32)
p = np.concatenate([c, bg])
run("tight blob + sparse background", p[:, 0], p[:, 1], p[:, 2])
g = np.stack(np.meshgrid(np.arange(60), np.arange(60), np.arange(60), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
run("integer lattice (mass ties)", g[:, 0], g[:, 1], g[:, 2])
line = np.stack([np.linspace(0, 1000, n), np.zeros(n), np.zeros(n)], 1).astype(np.float32)
run("collinear", line[:, 0], line[:, 1], line[:, 2])
