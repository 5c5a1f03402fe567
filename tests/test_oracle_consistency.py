"""The oracle's accelerated (grid) search must equal its brute-force definition bit for bit,
and its small solvers must satisfy their defining equations."""
import numpy as np
import pytest

import gen

F = np.float32


def _cloud(kind, n, seed):
    if kind == "uniform":
        return gen.uniform_cloud(n, seed)
    if kind == "hemisphere":
        return gen.hemisphere(n, seed, 5.0)
    if kind == "kitti":
        return gen.kitti_scene(seed, 3000, 300, 100)
    if kind == "degenerate":
        return gen.degenerate_cloud(max(n, 300), seed)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["uniform", "hemisphere", "kitti", "degenerate"])
@pytest.mark.parametrize("k", [1, 5, 17])
def test_grid_equals_bruteforce(oracle, kind, k):
    x, y, z = _cloud(kind, 2500, 11)
    i0, d0, q0, c0 = oracle.knn(x, y, z, x, y, z, k, algo=0)
    i1, d1, q1, c1 = oracle.knn(x, y, z, x, y, z, k, algo=1)
    assert np.array_equal(c0, c1)
    assert np.array_equal(i0, i1)
    assert np.array_equal(d0, d1, equal_nan=True)


def test_grid_equals_bruteforce_external_queries(oracle):
    x, y, z = gen.uniform_cloud(3000, 5, 0, 10)
    qx, qy, qz = gen.uniform_cloud(500, 6, -5, 15)  # many queries outside the grid
    a = oracle.knn(x, y, z, qx, qy, qz, 3, algo=0)
    b = oracle.knn(x, y, z, qx, qy, qz, 3, algo=1)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_knn_definition_small(oracle):
    """(d2 ascending, index descending) order against numpy on exactly the same f32 expression."""
    x, y, z = gen.degenerate_cloud(300, 3)
    k = 9
    idx, dist, d2, cnt = oracle.knn(x, y, z, x, y, z, k)
    valid = np.isfinite(x) & np.isfinite(y) & np.isfinite(z)
    for i in range(len(x)):
        if not valid[i]:
            assert cnt[i] == 0
            continue
        dx, dy, dz = (x[i] - x).astype(F), (y[i] - y).astype(F), (z[i] - z).astype(F)
        dd = ((dx * dx).astype(F) + (dy * dy).astype(F)).astype(F) + (dz * dz).astype(F)
        dd = dd.astype(F)
        cand = np.nonzero(valid)[0]
        order = cand[np.lexsort((-cand, dd[cand]))][:k]
        assert list(idx[i, : cnt[i]]) == list(order)
        assert np.array_equal(d2[i, : cnt[i]], dd[order])
        assert np.array_equal(dist[i, : cnt[i]], np.sqrt(dd[order]).astype(F))


def test_sor_mean_definition(oracle):
    x, y, z = gen.uniform_cloud(400, 9, 0, 5)
    k = 6
    _, dist, _, cnt = oracle.knn(x, y, z, x, y, z, k + 1)
    mean = oracle.sor_mean_dist(x, y, z, k)
    for i in range(len(x)):
        s = F(0)
        for j in range(1, cnt[i]):
            s = F(s + dist[i, j])
        assert mean[i] == F(s / F(cnt[i] - 1))


def test_sor_threshold_is_sequential_f32(oracle):
    rng = np.random.default_rng(0)
    m = rng.uniform(0.1, 3.0, 5000).astype(F)
    m[::97] = np.inf
    ok, thr, mu, sd = oracle.sor_threshold(m, 1.5)
    fin = m[np.isfinite(m)]
    s = F(0)
    for v in fin:
        s = F(s + v)
    mu_ref = F(s / F(len(fin)))
    v = F(0)
    for d in fin:
        e = F(d - mu_ref)
        v = F(v + F(e * e))
    sd_ref = np.sqrt(F(v / F(len(fin))))
    assert ok and mu == mu_ref and sd == sd_ref and thr == F(mu_ref + F(F(1.5) * sd_ref))


def test_eigenvector_is_smallest(oracle):
    rng = np.random.default_rng(1)
    for _ in range(200):
        pts = rng.normal(size=(12, 3)) * rng.uniform(0.1, 3, size=3)
        c = pts - pts.mean(0)
        A = (c.T @ c).astype(F)
        v = oracle.eigvec(A[0, 0], A[0, 1], A[0, 2], A[1, 1], A[1, 2], A[2, 2]).astype(np.float64)
        w, V = np.linalg.eigh(A.astype(np.float64))
        assert abs(np.linalg.norm(v) - 1) < 1e-5
        assert abs(abs(v @ V[:, 0]) - 1) < 1e-4 * max(1.0, w[2] / max(w[1] - w[0], 1e-12))


def test_solve_plane_satisfies_normal_equations(oracle):
    rng = np.random.default_rng(2)
    J = rng.normal(size=(50, 6))
    r = rng.normal(size=50) * 0.01
    ata, atb = J.T @ J, J.T @ r
    rot, t = oracle.solve_plane(ata.reshape(-1), atb)
    xs = np.linalg.solve(ata + 1e-6 * np.abs(np.diag(ata)).max() * np.eye(6), atb)
    assert np.allclose(t, xs[3:], rtol=1e-5, atol=1e-7)
    # Rodrigues of a small rotation vector: R - R^T = 2 sin(a)/a [w]x
    w = np.array([rot[2, 1] - rot[1, 2], rot[0, 2] - rot[2, 0], rot[1, 0] - rot[0, 1]]) / 2
    assert np.allclose(w, xs[:3], rtol=1e-3, atol=1e-6)
    assert np.allclose(rot @ rot.T, np.eye(3), atol=1e-5)


def test_p2p_recovers_rigid_motion(oracle):
    x, y, z = gen.uniform_cloud(200, 4, -1, 1)
    R = gen.rot_z(0.02)
    tx, ty, tz = oracle.apply_transform(x, y, z, R, (0.01, -0.02, 0.005))
    rc, res, _ = oracle.icp(0, (x, y, z), (tx, ty, tz), max_iter=30, tol=1e-9)
    assert rc == 0 and res.rmse < 1e-4
    assert np.allclose(np.array(res.rot).reshape(3, 3), R, atol=1e-4)


def test_voxel_downsample_against_dict_restatement(oracle):
    """Second, independent restatement of voxel_downsample.rs:24-62 (a dict of running f32 sums in
    input order, keys sorted at the end) against the sort-based oracle."""
    rng = np.random.default_rng(17)
    for n, size in ((1, 1.0), (500, 0.8), (3000, 3.3), (2000, 0.05)):
        pts = rng.uniform(-30, 30, size=(n, 3)).astype(np.float32)
        pts[rng.integers(0, n, size=max(n // 50, 1))] = np.nan
        f = np.float32
        bins = {}
        for px, py, pz in pts:
            if not (np.isfinite(px) and np.isfinite(py) and np.isfinite(pz)):
                continue
            key = tuple(int(np.floor(f(v) / f(size))) for v in (px, py, pz))
            a = bins.setdefault(key, [f(0), f(0), f(0), 0])
            a[0] = f(a[0] + px)
            a[1] = f(a[1] + py)
            a[2] = f(a[2] + pz)
            a[3] += 1
        keys = sorted(bins)
        ref = np.array([[f(bins[k][i] / f(bins[k][3])) for i in range(3)] for k in keys],
                       dtype=np.float32).reshape(-1, 3)
        got = oracle.voxel_downsample(pts[:, 0], pts[:, 1], pts[:, 2], size)
        assert len(got[0]) == len(keys)
        for i in range(3):
            assert np.array_equal(got[i].view(np.uint32), ref[:, i].copy().view(np.uint32))


def test_euclidean_cluster_against_dense_graph(oracle):
    """The cell-restricted restatement against connected components of the dense
    "distance <= threshold" graph (scipy), same f32 distance expression."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components

    rng = np.random.default_rng(23)
    f = np.float32
    for n, thr in ((300, 1.5), (800, 0.9), (500, 4.0)):
        p = rng.uniform(-10, 10, size=(n, 3)).astype(f)
        d = p[:, None, :] - p[None, :, :]
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        adj = csr_matrix(d2 <= f(thr) * f(thr))
        ncomp, lab = connected_components(adj, directed=False)
        comps = [np.nonzero(lab == c)[0] for c in range(ncomp)]
        comps = [c for c in comps if 2 <= len(c) <= 200]
        comps.sort(key=lambda c: (-len(c), c[0]))
        got = oracle.euclidean_cluster(p[:, 0], p[:, 1], p[:, 2], thr, 2, 200)
        assert len(got) == len(comps)
        for a, b in zip(got, comps):
            assert np.array_equal(a, b)
