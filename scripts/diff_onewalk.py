#!/usr/bin/env python3
"""Differential run of the two lane-per-query kernels that return neighbour identities: prints one
checksum line per (cloud, k) over k-NN lists, distances and counts of many seeded clouds (sizes
and shapes the CPU oracle could not cover in reasonable time).  Run it once as it is and once with
PCRS_ONE_WALK=0 and compare the outputs: every line must be identical."""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))
import numpy as np
from pointclouds_rs import hip

def clouds():
    rng = np.random.default_rng(77)
    for n in (50_000, 400_000, 2_000_000):
        u = rng.uniform(0, 100, size=(n, 3))
        yield f"uniform{n}", u
        for q in (0.02, 0.2, 1.0):
            yield f"uniform{n}_q{q}", np.round(u / q) * q
        c = rng.normal(size=(n, 3)) * rng.choice([0.05, 0.5, 5.0], size=(n, 1)) + rng.integers(0, 6, size=(n, 3)) * 20.0
        yield f"blobs{n}", c
        p = np.stack([rng.uniform(0, 200, n), rng.uniform(0, 200, n), 0.01 * rng.normal(size=n)], 1)
        yield f"plane{n}", p
        yield f"plane{n}_q", np.round(p, 1)
        r = 1.0 + 60.0 * rng.uniform(size=n) ** 2
        a = rng.uniform(0, 2 * np.pi, n)
        yield f"rings{n}", np.stack([r * np.cos(a), r * np.sin(a), 0.02 * rng.normal(size=n)], 1)
        d = np.repeat(rng.uniform(0, 50, size=(n // 4, 3)), 4, axis=0)
        yield f"quadruples{n}", d
    g = np.stack(np.meshgrid(np.arange(80), np.arange(80), np.arange(80), indexing="ij"), -1).reshape(-1, 3)
    yield "lattice80", g * 0.37
    yield "line", np.stack([np.linspace(0, 1000, 300_000), np.zeros(300_000), np.zeros(300_000)], 1)

for name, p in clouds():
    p = p.astype(np.float32)
    x, y, z = (np.ascontiguousarray(p[:, i]) for i in range(3))
    for k in (3, 8, 16, 17, 24):
        ix = hip.Index(x, y, z, k_hint=k, algo=2)
        nq = min(len(x), 300_000)
        idx, dist, cnt = ix.knn(x[:nq], y[:nq], z[:nq], k)
        st = ix.last_stats(); ix.close()
        crc = zlib.crc32(dist.tobytes(), zlib.crc32(idx.tobytes(), zlib.crc32(cnt.tobytes())))
        print(f"{name} k={k} crc={crc:08x} answered={sum(st.values())}", flush=True)
    keep, nk, thr, nrm = hip.sor_normals(x, y, z, 16, 1.0, 16)
    crc = zlib.crc32(keep.tobytes())
    for a in nrm: crc = zlib.crc32(np.ascontiguousarray(a).tobytes(), crc)
    print(f"{name} sor+normals crc={crc:08x} kept={nk} thr={np.float32(thr).view(np.uint32):08x}", flush=True)
