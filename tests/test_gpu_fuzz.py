"""Randomised parity sweep: many small clouds of awkward shapes, random k, every consumer, all
through the C-ABI against the oracle (bit-exact except normals, 1e-5)."""
import numpy as np
import pytest

import gen

F = np.float32
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from pointclouds_rs import hip as h

    return h


def make_cloud(rng, kind, n):
    if kind == "uniform":
        p = rng.uniform(-10, 10, size=(n, 3))
    elif kind == "clusters":
        c = rng.uniform(-50, 50, size=(max(n // 40, 1), 3))
        p = c[rng.integers(0, len(c), n)] + rng.normal(0, 0.05, size=(n, 3))
    elif kind == "lattice":
        p = rng.integers(0, 7, size=(n, 3)).astype(np.float64) * 0.5
    elif kind == "plane":
        p = np.concatenate([rng.uniform(-5, 5, size=(n, 2)), np.zeros((n, 1))], 1)
    elif kind == "line":
        p = np.concatenate([rng.uniform(-5, 5, size=(n, 1)), np.zeros((n, 2))], 1)
    elif kind == "duplicates":
        base = rng.uniform(-3, 3, size=(max(n // 10, 1), 3))
        p = base[rng.integers(0, len(base), n)]
    else:  # "wide": a dense core and a few points very far away
        p = rng.normal(0, 1, size=(n, 3))
        far = rng.integers(0, n, max(n // 50, 1))
        p[far] *= 1e6
    p = p.astype(F)
    bad = rng.random(n) < 0.02
    p[bad, rng.integers(0, 3, int(bad.sum()))] = rng.choice([np.nan, np.inf, -np.inf], int(bad.sum()))
    return p[:, 0].copy(), p[:, 1].copy(), p[:, 2].copy()


KINDS = ["uniform", "clusters", "lattice", "plane", "line", "duplicates", "wide"]


@pytest.mark.parametrize("seed", range(16))
@pytest.mark.parametrize("kind", KINDS)
def test_random_clouds(hip, oracle, kind, seed):
    rng = np.random.default_rng(1000 * KINDS.index(kind) + seed)
    n = int(rng.choice([1, 2, 3, 17, 100, 1000, 5000, 9000]))
    k = int(rng.choice([1, 2, 5, 10, 16, 17, 24, 33, 64, 70]))
    x, y, z = make_cloud(rng, kind, n)
    algo = 2 if n >= 64 else 0
    # knn lists over the cloud itself and over a few outside queries
    q = np.concatenate([np.stack([x, y, z], 1)[: min(n, 300)], rng.uniform(-60, 60, size=(20, 3)).astype(F)])
    ix = hip.Index(x, y, z, k_hint=k, algo=algo)
    idx, dist, cnt = ix.knn(q[:, 0], q[:, 1], q[:, 2], k)
    ridx, rdist, _, rcnt = oracle.knn(x, y, z, q[:, 0], q[:, 1], q[:, 2], k, algo=0)
    assert np.array_equal(cnt, rcnt) and np.array_equal(idx, ridx)
    assert np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
    rr = float(rng.choice([0.05, 0.4, 2.0]))
    off, lst = ix.radius_search(q[:, 0], q[:, 1], q[:, 2], rr)
    roff, rlst = oracle.radius_search(x, y, z, q[:, 0], q[:, 1], q[:, 2], rr)
    assert np.array_equal(off, roff) and np.array_equal(lst, rlst)
    t, d, nc = ix.find_correspondences(q[:, 0], q[:, 1], q[:, 2], 5.0)
    rt, rd, rnc = oracle.find_correspondences(q[:, 0], q[:, 1], q[:, 2], x, y, z, 5.0, algo=0)
    assert nc == rnc and np.array_equal(t, rt) and np.array_equal(d.view(np.uint32), rd.view(np.uint32))
    ix.close()
    # SOR and normals
    ks = min(k, 40)
    keep, n_keep, thr, mean = hip.sor(x, y, z, ks, 1.5)
    rkeep, rn, rthr, rmean = oracle.sor(x, y, z, ks, 1.5, algo=0)
    assert np.array_equal(mean.view(np.uint32), rmean.view(np.uint32))
    assert np.array_equal(keep, rkeep) and n_keep == rn
    assert (np.isnan(thr) and np.isnan(rthr)) or F(thr) == F(rthr)
    vp = (1.0, -2.0, 30.0)
    nrm = hip.normals(x, y, z, ks, vp)
    rnrm = oracle.normals(x, y, z, ks, vp, algo=0)
    if kind not in ("lattice", "line", "duplicates", "plane"):
        for a, b in zip(nrm, rnrm):
            assert np.abs(a.astype(np.float64) - b).max(initial=0.0) <= 1e-5
    else:
        # Degenerate covariances: the eigenvector of the smallest eigenvalue is not unique there,
        # so the components are not compared.  What every answer must still be: finite, of unit
        # length, turned towards the viewpoint (estimate.rs:100-108), the default where the
        # oracle has the default, and as good a minimiser of n^T C n as the oracle's own normal.
        N = np.stack(nrm, 1).astype(np.float64)
        RN = np.stack(rnrm, 1).astype(np.float64)
        P = np.stack([x, y, z], 1).astype(np.float64)
        assert np.isfinite(N).all()
        assert np.abs(np.linalg.norm(N, axis=1) - 1.0).max(initial=0.0) <= 1e-5
        default = (RN == np.array([0.0, 0.0, 1.0])).all(1)
        assert np.array_equal(N[default], RN[default])
        fin = np.isfinite(P).all(1) & ~default
        toward = ((np.array(vp) - P[fin]) * N[fin]).sum(1)
        assert (toward >= -1e-4 * (1.0 + np.abs(P[fin]).max(initial=0.0))).all()
        nidx, _, _, ncnt = oracle.knn(x, y, z, x, y, z, ks, algo=0)
        for i in np.flatnonzero(fin)[:400]:
            nb = P[nidx[i, : ncnt[i]]]
            C = (nb - nb.mean(0)).T @ (nb - nb.mean(0))
            scale = np.trace(C) + 1e-30
            assert N[i] @ C @ N[i] <= RN[i] @ C @ RN[i] + 1e-4 * scale
    # the stages next to the path
    r = float(rng.choice([0.05, 0.4, 2.0]))
    _, _, count = hip.radius_outlier(x, y, z, r, 3)
    _, _, rcount = oracle.radius_outlier(x, y, z, r, 3)
    assert np.array_equal(count, rcount)
    v = hip.voxel_downsample(x, y, z, r)
    rv = oracle.voxel_downsample(x, y, z, r)
    for a, b in zip(v, rv):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    cl = hip.euclidean_cluster(x, y, z, r, 2, 500)
    rcl = oracle.euclidean_cluster(x, y, z, r, 2, 500)
    assert len(cl) == len(rcl) and all(np.array_equal(a, b) for a, b in zip(cl, rcl))


@pytest.mark.parametrize("seed", range(24))
def test_random_uneven_clouds(hip, oracle, seed):
    """Clouds large and uneven enough for the index to grow further grids: a sparse field with a
    random handful of dense clusters, patches of plane and lines of very different sizes and
    densities (some of them worth a finer grid of their own, some not, some nested in another's
    box), a few non-finite points, random k.  SOR means, threshold and mask bit-exact against the
    oracle; normals within 1e-5 away from the degenerate parts."""
    rng = np.random.default_rng(7000 + seed)
    half = float(rng.choice([20.0, 50.0, 200.0]))
    parts = [rng.uniform(-half, half, size=(int(rng.integers(60_000, 120_000)), 3))]
    clean = [np.ones(len(parts[0]), dtype=bool)]
    for _ in range(int(rng.integers(2, 7))):
        c = rng.uniform(-0.8 * half, 0.8 * half, size=3)
        m = int(rng.choice([3_000, 12_000, 25_000, 40_000]))
        kind = rng.choice(["ball", "plane", "line"])
        sigma = half * float(rng.choice([2e-4, 2e-3, 1e-2]))
        if kind == "ball":
            p = rng.normal(0, sigma, size=(m, 3)) + c
        elif kind == "plane":
            p = np.concatenate([rng.uniform(-30 * sigma, 30 * sigma, size=(m, 2)),
                                rng.normal(0, 0.02 * sigma, size=(m, 1))], 1) + c
        else:
            p = np.concatenate([rng.uniform(-100 * sigma, 100 * sigma, size=(m, 1)),
                                rng.normal(0, 0.02 * sigma, size=(m, 2))], 1) + c
        parts.append(p)
        clean.append(np.full(m, kind == "ball"))
    p = np.concatenate(parts).astype(F)
    clean = np.concatenate(clean)
    order = rng.permutation(len(p))
    p, clean = p[order], clean[order]
    bad = rng.random(len(p)) < 1e-3
    p[bad, rng.integers(0, 3, int(bad.sum()))] = rng.choice([np.nan, np.inf, -np.inf], int(bad.sum()))
    x, y, z = (np.ascontiguousarray(p[:, a]) for a in range(3))
    k = int(rng.choice([5, 10, 16, 24]))
    vp = (0.0, 0.0, 10.0 * half)
    rkeep, rn, rthr, rmean = oracle.sor(x, y, z, k, 1.0, algo=1)
    ref = oracle.normals(x, y, z, k, vp, algo=1)
    for _ in range(2):  # the second build starts every grid from what the first ended with
        keep, n_keep, thr, mean = hip.sor(x, y, z, k, 1.0)
        badm = np.flatnonzero(mean.view(np.uint32) != rmean.view(np.uint32))
        assert len(badm) == 0, (seed, k, len(badm), badm[:5], mean[badm[:5]], rmean[badm[:5]])
        assert (F(thr) == F(rthr) or (np.isnan(thr) and np.isnan(rthr))) and n_keep == rn
        assert np.array_equal(keep, rkeep)
        keep2, nk2, thr2, nrm = hip.sor_normals(x, y, z, k, 1.0, k, vp)
        assert nk2 == rn and np.array_equal(keep2, rkeep)
        ok = np.ones(len(x), dtype=bool)
        for a, b in zip(nrm, ref):
            ok &= np.abs(a.astype(np.float64) - b) <= 1e-5
        assert ok[clean & ~bad].mean() > 0.9999, (seed, k, ok[clean & ~bad].mean())


@pytest.mark.parametrize("seed", range(5))
@pytest.mark.parametrize("kind", KINDS)
def test_random_clouds_runs_of_cells(kind, seed):
    """Ownership by runs of cells on the same awkward shapes (finite points only: a cloud with
    non-finite points has no cell order to split): the runs partition the cloud and every answer
    equals the answer of the index-range entry point for the point it names, bit for bit."""
    from pointclouds_rs.dist import HipEngine

    eng = HipEngine.shared(0)
    torch = eng.torch
    rng = np.random.default_rng(7000 + 100 * KINDS.index(kind) + seed)
    n = int(rng.choice([64, 100, 1000, 5000, 9000]))
    k = int(rng.choice([1, 5, 10, 16, 24, 33]))
    parts = int(rng.choice([1, 2, 3, 5, 8]))
    cols_h = [np.nan_to_num(c, nan=0.25, posinf=1.5, neginf=-1.5) for c in make_cloud(rng, kind, n)]
    cols = [eng.from_numpy(c) for c in cols_h]
    vp = (1.0, -2.0, 30.0)
    ix = eng.build_index(*cols, k + 1, 2)
    ref = [eng.empty(n) for _ in range(4)]
    eng.sor_normals_range(ix, 0, n, k, k, vp, *ref)
    bounds = eng.slot_bounds(ix, parts)
    if bounds[-1] == 0:  # no grid for this cloud (coincident points): nothing to split
        eng.destroy_index(ix)
        return
    assert bounds[0] == 0 and bounds[-1] == n and bounds == sorted(bounds)
    seen = np.zeros(n, dtype=np.int64)
    for r in range(parts):
        b, c = bounds[r], bounds[r + 1] - bounds[r]
        point = eng.empty(c, "int32")
        out = [eng.empty(c) for _ in range(4)]
        eng.sor_normals_slots(ix, b, c, k, k, vp, point, *out)
        torch.cuda.synchronize()
        pt = point.cpu().numpy().view(np.uint32)
        seen[pt] += 1
        for got, want in zip(out, ref):
            assert np.array_equal(got.cpu().numpy().view(np.uint32), want.cpu().numpy()[pt].view(np.uint32))
    assert (seen == 1).all()
    eng.destroy_index(ix)
