#!/usr/bin/env python3
"""Experiment: one GPU, a cloud of argv[1] points (default 50 M): fused SOR k=16 + normals k=16,
timed, and checked on a random sample of queries against the brute-force kernel of the same
library (an independent code path: LDS-tiled scan of all points)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np, torch
from pointclouds_rs import _ffi, synth
from pointclouds_rs.dist import HipEngine

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 50_000_000
eng = HipEngine(0)
side = np.float32(100.0 * (n / 1e6) ** (1.0 / 3.0))
cols = [eng.from_numpy((synth.uniform01(5, n, a) * side).astype(np.float32)) for a in range(3)]
mean = eng.empty(n); nx = eng.empty(n); ny = eng.empty(n); nz = eng.empty(n); keep = eng.empty(n, "uint8")
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ix = eng.build_index(*cols, 17)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    eng.sor_normals_range(ix, 0, n, 16, 16, (0.0, 0.0, 0.0), mean, nx, ny, nz)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    thr, ok, n_keep = eng.select(mean, 1.0, mean, keep)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    st = eng.index_stats(ix)
    eng.destroy_index(ix)
    print(f"n={n} build {1e3*(t1-t0):.2f} ms  search {1e3*(t2-t1):.2f} ms  stats+mask {1e3*(t3-t2):.2f} ms "
          f"-> {n/(t3-t0)/1e6:.0f} Mpoints/s  thr={thr:.6f} kept={n_keep} stages={st}", flush=True)
# sample check against the brute-force kernel
rng = np.random.default_rng(1)
sample = np.sort(rng.choice(n, size=2048, replace=False))
st = torch.from_numpy(sample).to(eng.device)
q = [c[st].contiguous() for c in cols]
bf = C.c_void_p()
eng.ctx.check(eng.lib.pcrs_index_build_dev(eng.ctx.handle, *[eng._p(c) for c in cols], n, 17, _ffi.ALGO_BRUTE, C.byref(bf)))
k = 17
idx = eng.empty(2048 * k, "int32"); dist = eng.empty(2048 * k); cnt = eng.empty(2048, "int32")
t0 = time.perf_counter()
eng.ctx.check(eng.lib.pcrs_knn_batch_dev(bf, *[eng._p(c) for c in q], 2048, k, eng._p(idx), eng._p(dist), eng._p(cnt)))
torch.cuda.synchronize()
print(f"brute force over {n} points for 2048 queries: {1e3*(time.perf_counter()-t0):.1f} ms")
eng.lib.pcrs_index_destroy(bf)
d = dist.cpu().numpy().reshape(2048, k)
s = np.zeros(2048, dtype=np.float32)
for j in range(1, k):
    s = (s + d[:, j]).astype(np.float32)
ref = (s / np.float32(16)).astype(np.float32)
got = mean.cpu().numpy()[sample]
bad = int((got.view(np.uint32) != ref.view(np.uint32)).sum())
print("sample means bit-exact:", bad == 0, "mismatches", bad)
sys.exit(0 if bad == 0 else 1)
