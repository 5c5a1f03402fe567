import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import gen
import oracle_ffi
from pointclouds_rs.dist import HipEngine
oracle = oracle_ffi.load()
eng = HipEngine(0)
x, y, z = gen.degenerate_cloud(300, 7)
ux, uy, uz = gen.uniform_cloud(30000, 11)
x, y, z = np.concatenate([x, ux]), np.concatenate([y, uy]), np.concatenate([z, uz])
n = len(x)
cols = [eng.from_numpy(c) for c in (x, y, z)]
rn = oracle.normals(x, y, z, 16, (1.0, 2.0, 3.0), algo=1)
ridx, rdist, rd2, rcnt = oracle.knn(x, y, z, x, y, z, 16, algo=1)
for rep in range(3):
    ix = eng.build_index(*cols, 17, 2)
    mean = eng.empty(n); nx = eng.empty(n); ny = eng.empty(n); nz = eng.empty(n)
    eng.sor_normals_range(ix, 0, n, 16, 16, (1.0, 2.0, 3.0), mean, nx, ny, nz)
    eng.torch.cuda.synchronize()
    g = np.stack([t.cpu().numpy() for t in (nx, ny, nz)], 1)
    r = np.stack(rn, 1)
    bad = np.where(np.abs(g.astype(np.float64) - r).max(1) > 1e-5)[0]
    print("rep", rep, "bad rows", len(bad), bad[:10], eng.index_stats(ix))
    for b in bad[:3]:
        print(" row", b, "pt", x[b], y[b], z[b], "got", g[b], "ref", r[b])
        print("  ref idx", ridx[b], "d2", rd2[b].view(np.uint32))
    eng.destroy_index(ix)
