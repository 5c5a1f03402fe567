"""GPU parity: libpcrs_hip.so (through the C-ABI) against the CPU oracle on the same seeded
inputs.  Bit-exact for neighbour indices, squared/Euclidean distances, SOR means, thresholds and
masks; 1e-5 for normals (f64 acos/cos differ between the device and host math libraries)."""
import numpy as np
import pytest

import gen

pytestmark = pytest.mark.gpu
F = np.float32
NO = 0xFFFFFFFF


@pytest.fixture(scope="module")
def hip():
    from pointclouds_rs import hip as h

    return h


def clouds():
    return {
        "uniform3k": gen.uniform_cloud(3000, 42),
        "hemisphere2k": gen.hemisphere(2000, 99, 5.0),
        "kitti": gen.kitti_scene(42, 4000, 300, 100),
        "degenerate": gen.degenerate_cloud(300, 7),
        "tiny": gen.uniform_cloud(5, 3),
    }


ALGOS = [1, 2]  # PCRS_ALGO_BRUTE, PCRS_ALGO_GRID


@pytest.mark.parametrize("algo", ALGOS, ids=["brute", "grid"])
@pytest.mark.parametrize("name", list(clouds()))
@pytest.mark.parametrize("k", [1, 4, 10, 17, 33, 70, 200])
def test_knn_self_bit_exact(hip, oracle, name, k, algo):
    x, y, z = clouds()[name]
    ix = hip.Index(x, y, z, k_hint=k, algo=algo)
    idx, dist, cnt = ix.knn(x, y, z, k)
    ix.close()
    ridx, rdist, _, rcnt = oracle.knn(x, y, z, x, y, z, k, algo=1)
    assert np.array_equal(cnt, rcnt)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))


@pytest.mark.parametrize("algo", ALGOS, ids=["brute", "grid"])
def test_knn_external_queries(hip, oracle, algo):
    x, y, z = gen.uniform_cloud(6000, 5, 0, 10)
    qx, qy, qz = gen.uniform_cloud(1500, 6, -5, 15)  # many queries outside the cloud's box
    qx = qx.copy()
    qx[7] = np.nan
    qy[11] = np.inf
    for k in (1, 3, 16):
        ix = hip.Index(x, y, z, k_hint=k, algo=algo)
        idx, dist, cnt = ix.knn(qx, qy, qz, k)
        ix.close()
        ridx, rdist, _, rcnt = oracle.knn(x, y, z, qx, qy, qz, k, algo=1)
        assert np.array_equal(cnt, rcnt) and cnt[7] == 0 and cnt[11] == 0
        assert np.array_equal(idx, ridx)
        assert np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))


@pytest.mark.parametrize("n,k", [(20000, 10), (50000, 16)])
def test_sor_parity_mid_size(hip, oracle, n, k):
    x, y, z = gen.uniform_cloud(n, 42)
    keep, n_keep, thr, mean = hip.sor(x, y, z, k, 1.0)
    rkeep, rn, rthr, rmean = oracle.sor(x, y, z, k, 1.0, algo=1)
    assert np.array_equal(mean.view(np.uint32), rmean.view(np.uint32))
    assert F(thr).view(np.uint32) == F(rthr).view(np.uint32)
    assert n_keep == rn and np.array_equal(keep, rkeep)
    assert 0 < n_keep < n


@pytest.mark.parametrize("name", ["degenerate", "kitti", "tiny", "hemisphere2k"])
def test_sor_parity_irregular(hip, oracle, name):
    x, y, z = clouds()[name]
    for k, mul in ((1, 0.0), (8, 1.0), (20, 2.0), (400, 1.0)):
        keep, n_keep, thr, mean = hip.sor(x, y, z, k, mul)
        rkeep, rn, rthr, rmean = oracle.sor(x, y, z, k, mul, algo=0)
        assert np.array_equal(mean.view(np.uint32), rmean.view(np.uint32)), (name, k)
        assert np.array_equal(keep, rkeep) and n_keep == rn
        assert F(thr).view(np.uint32) == F(rthr).view(np.uint32)


def test_sor_edge_cases(hip):
    e = np.empty(0, dtype=F)
    keep, n_keep, thr, _ = hip.sor(e, e, e, 5, 1.0)
    assert len(keep) == 0 and n_keep == 0
    keep, n_keep, _, _ = hip.sor([1], [2], [3], 5, 1.0)  # statistical_outlier.rs:10-12
    assert list(keep) == [True] and n_keep == 1
    keep, n_keep, _, _ = hip.sor([1, 2], [3, 4], [5, 6], 0, 1.0)  # :5-7
    assert not keep.any() and n_keep == 0
    nan = float("nan")
    keep, n_keep, thr, mean = hip.sor([nan, nan], [0, 0], [0, 0], 3, 1.0)  # :49-51
    assert not keep.any() and np.isnan(thr) and np.all(np.isinf(mean))
    # one finite point among non-finite ones: its only neighbour is itself, mean 0
    keep, n_keep, thr, mean = hip.sor([1, nan, nan], [1, 0, 0], [1, 0, 0], 3, 1.0)
    assert list(keep) == [True, False, False] and mean[0] == 0.0


def normals_close(got, ref, tol):
    g = np.stack(got, 1).astype(np.float64)
    r = np.stack(ref, 1).astype(np.float64)
    return np.abs(g - r).max(axis=1) <= tol


@pytest.mark.parametrize("name,k", [("uniform3k", 16), ("hemisphere2k", 15), ("kitti", 10),
                                    ("degenerate", 8), ("tiny", 3), ("uniform3k", 100)])
def test_normals_parity(hip, oracle, name, k):
    x, y, z = clouds()[name]
    vp = (0.5, -1.0, 30.0)
    got = hip.normals(x, y, z, k, vp)
    ref = oracle.normals(x, y, z, k, vp, algo=0)
    ok = normals_close(got, ref, 1e-5)
    exact = np.all(np.stack(got, 1).view(np.uint32) == np.stack(ref, 1).view(np.uint32), axis=1)
    # A flipped sign is legitimate only where n . (viewpoint - p) rounds to zero.
    bad = np.nonzero(~ok)[0]
    for i in bad:
        flipped = normals_close(tuple(-c[i:i + 1] for c in got), tuple(c[i:i + 1] for c in ref), 1e-5)[0]
        v = np.array(vp) - np.array([x[i], y[i], z[i]], dtype=np.float64)
        dot = abs(float(np.dot([c[i] for c in ref], v)))
        assert flipped and dot < 1e-4 * np.linalg.norm(v), (name, i, [c[i] for c in got], [c[i] for c in ref])
    assert exact.mean() > 0.95, f"only {exact.mean():.3f} bit-exact"


def test_normals_50k_parity(hip, oracle):
    x, y, z = gen.uniform_cloud(50000, 42)
    got = hip.normals(x, y, z, 16)
    ref = oracle.normals(x, y, z, 16, algo=1)
    ok = normals_close(got, ref, 1e-5)
    assert ok.mean() > 0.9999, ok.mean()


def test_sor_normals_fused_matches_separate(hip):
    x, y, z = gen.uniform_cloud(20000, 8)
    keep, n_keep, thr, nrm = hip.sor_normals(x, y, z, 16, 1.0, 16)
    keep2, n_keep2, thr2, _ = hip.sor(x, y, z, 16, 1.0)
    nrm2 = hip.normals(x, y, z, 16)
    assert np.array_equal(keep, keep2) and n_keep == n_keep2 and thr == thr2
    for a, b in zip(nrm, nrm2):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_radius_count_parity(hip, oracle):
    x, y, z = gen.degenerate_cloud(400, 2)
    ix = hip.Index(x, y, z)
    for r in (0.3, 1.0, 2.5):
        c = ix.radius_count(x, y, z, r)
        _, _, rc = oracle.radius_outlier(x, y, z, r, 1)
        assert np.array_equal(c, rc)
    assert not ix.radius_count(x, y, z, -1.0).any()
    assert not ix.radius_count(x, y, z, float("nan")).any()
    ix.close()


def test_radius_count_grid_parity(hip, oracle):
    """Grid-backed radius counts: cloud queries, queries outside the bounding box, radii from a
    fraction of a cell to several cells (the largest falls back to the brute-force scan)."""
    ux, uy, uz = gen.uniform_cloud(20000, 5, 0, 10)
    dx, dy, dz = gen.degenerate_cloud(300, 9)
    x, y, z = np.concatenate([ux, dx]), np.concatenate([uy, dy]), np.concatenate([uz, dz])
    qx = np.concatenate([x[:3000], np.array([-5, 15, 5, np.nan], dtype=F)])
    qy = np.concatenate([y[:3000], np.array([5, 5, -20, 1], dtype=F)])
    qz = np.concatenate([z[:3000], np.array([5, 12, 5, 1], dtype=F)])
    ix = hip.Index(x, y, z, k_hint=16, algo=2)
    for r in (0.05, 0.4, 0.9, 1.7, 6.0):
        c = ix.radius_count(qx, qy, qz, r)
        d2 = ((qx[:, None].astype(F) - x[None, :]) ** 2 + (qy[:, None] - y[None, :]) ** 2) \
            + (qz[:, None] - z[None, :]) ** 2
        ref = (d2 <= F(r) * F(r)).sum(1).astype(np.uint32)  # NaN compares false
        assert np.array_equal(c, ref), r
    ix.close()


def test_radius_search_lists_parity(hip, oracle):
    """Batched KdTree::radius_search as CSR lists, identical to the oracle's: through the grid
    (lists ordered by a key sort) and through the brute-force scan, cloud points and outside
    points as queries, non-finite queries and points, unusable radii, lists that do not fit."""
    ux, uy, uz = gen.uniform_cloud(20000, 5, 0, 10)
    dx, dy, dz = gen.degenerate_cloud(300, 9)
    x, y, z = np.concatenate([ux, dx]), np.concatenate([uy, dy]), np.concatenate([uz, dz])
    qx = np.concatenate([x[:2000], np.array([-5, 15, 5, np.nan], dtype=F)])
    qy = np.concatenate([y[:2000], np.array([5, 5, -20, 1], dtype=F)])
    qz = np.concatenate([z[:2000], np.array([5, 12, 5, 1], dtype=F)])
    ix = hip.Index(x, y, z, k_hint=16, algo=2)
    for r in (0.05, 0.4, 0.9, 1.7, 6.0):  # the largest is answered by the brute-force scan
        off, idx = ix.radius_search(qx, qy, qz, r)
        roff, ridx = oracle.radius_search(x, y, z, qx, qy, qz, r)
        assert np.array_equal(off, roff), r
        assert np.array_equal(idx, ridx), r
        assert np.array_equal(np.diff(off.astype(np.int64)), ix.radius_count(qx, qy, qz, r))
    for r in (-1.0, 0.0, float("nan"), float("inf")):
        off, idx = ix.radius_search(qx, qy, qz, r)
        assert not off.any() and len(idx) == 0
    off, idx = ix.radius_search([], [], [], 1.0)
    assert list(off) == [0] and len(idx) == 0
    # a list that does not fit: the call reports the sizes and fails without writing
    import ctypes as C

    from pointclouds_rs import _ffi

    n = len(qx)
    offsets = np.zeros(n + 1, dtype=np.uint32)
    small = np.full(8, 0xABCDABCD, dtype=np.uint32)
    total = C.c_size_t(0)
    rc = ix.ctx.lib.pcrs_radius_search(ix.handle, qx.ctypes.data_as(C.c_void_p),
                                       qy.ctypes.data_as(C.c_void_p), qz.ctypes.data_as(C.c_void_p),
                                       n, 0.9, offsets.ctypes.data_as(C.c_void_p),
                                       small.ctypes.data_as(C.c_void_p), len(small), C.byref(total))
    roff, ridx = oracle.radius_search(x, y, z, qx, qy, qz, 0.9)
    assert rc == _ffi.PCRS_ERR_INVALID_ARG and total.value == len(ridx)
    assert np.array_equal(offsets, roff) and (small == 0xABCDABCD).all()
    ix.close()
    # a small cloud (brute-force index) and an empty one
    sx, sy, sz = gen.degenerate_cloud(400, 2)
    ix = hip.Index(sx, sy, sz)
    for r in (0.3, 1.0, 2.5):
        off, idx = ix.radius_search(sx, sy, sz, r)
        roff, ridx = oracle.radius_search(sx, sy, sz, sx, sy, sz, r)
        assert np.array_equal(off, roff) and np.array_equal(idx, ridx)
    ix.close()
    ix = hip.Index([], [], [])
    off, idx = ix.radius_search([1.0], [2.0], [3.0], 5.0)
    assert list(off) == [0, 0] and len(idx) == 0
    ix.close()


@pytest.mark.parametrize("n", [500, 30000], ids=["brute", "grid"])
def test_radius_outlier_parity(hip, oracle, n):
    ux, uy, uz = gen.uniform_cloud(n, 21, 0, 20)
    dx, dy, dz = gen.degenerate_cloud(300, 3)
    x, y, z = np.concatenate([ux, dx]), np.concatenate([uy, dy]), np.concatenate([uz, dz])
    for r, mn in ((0.8, 4), (1.5, 1), (0.02, 2), (50.0, 10), (1.0, 0), (-1.0, 1), (float("nan"), 0),
                  (float("inf"), 1)):
        keep, n_keep, count = hip.radius_outlier(x, y, z, r, mn)
        rkeep, rn, rcount = oracle.radius_outlier(x, y, z, r, mn)
        assert np.array_equal(count, rcount), (r, mn)
        assert np.array_equal(keep, rkeep) and n_keep == rn, (r, mn)
    keep, n_keep, count = hip.radius_outlier([], [], [], 1.0, 1)
    assert len(keep) == 0 and n_keep == 0


def test_voxel_downsample_parity(hip, oracle):
    """Centroids bit-identical to the CPU restatement: key order, per-voxel summation order,
    negative coordinates, saturating keys, non-finite rows, one voxel holding everything."""
    ux, uy, uz = gen.uniform_cloud(50000, 13, -40, 60)
    dx, dy, dz = gen.degenerate_cloud(300, 5)
    x, y, z = np.concatenate([ux, dx]), np.concatenate([uy, dy]), np.concatenate([uz, dz])
    x = np.concatenate([x, np.array([3e9, -3e9, 1e20, 0.5], dtype=F)])
    y = np.concatenate([y, np.array([1.0, 2.0, -1e20, 0.25], dtype=F)])
    z = np.concatenate([z, np.array([0.0, 0.0, 7.0, -0.0], dtype=F)])
    for size in (0.07, 0.5, 2.0, 33.3, 1e6):
        got = hip.voxel_downsample(x, y, z, size)
        ref = oracle.voxel_downsample(x, y, z, size)
        assert len(got[0]) == len(ref[0]), size
        for a, b in zip(got, ref):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), size
    nan = np.full(5, np.nan, dtype=F)
    assert len(hip.voxel_downsample(nan, nan, nan, 1.0)[0]) == 0
    for bad in (0.0, -1.0, float("nan"), float("inf")):
        with pytest.raises(Exception):
            hip.voxel_downsample(x, y, z, bad)


@pytest.mark.parametrize("n,ns", [(3000, 200), (60000, 400), (60000, 8)],
                         ids=["sequential", "parallel", "few-samples"])
def test_ransac_plane_parity(hip, oracle, n, ns):
    """Same triples in, same plane / inliers / count out: both selection rules of the reference
    (early-stopping scan below 10000 points or 16 samples, earliest maximum above)."""
    import pointclouds_rs as pc

    rng = np.random.default_rng(n + ns)
    xy = rng.uniform(-20, 20, size=(n, 2)).astype(F)
    zc = (0.02 * xy[:, 0] - 0.01 * xy[:, 1] + 1.5 + rng.normal(0, 0.01, n)).astype(F)
    out = rng.random(n) < 0.3
    zc[out] = rng.uniform(-10, 10, int(out.sum())).astype(F)
    x, y, z = xy[:, 0].copy(), xy[:, 1].copy(), zc
    x[5], z[9] = np.nan, np.inf
    samples = pc.sample_triples(n, ns, 7)
    samples[3] = (1, 1, 2)  # coincident points: no plane
    for thr in (0.03, 0.5, 0.0):
        got = hip.ransac_plane_samples(x, y, z, samples, thr)
        ref = oracle.ransac_plane_samples(x, y, z, samples, thr)
        assert np.array_equal(got[0].view(np.uint32), ref[0].view(np.uint32)), thr
        assert np.float32(got[1]).view(np.uint32) == np.float32(ref[1]).view(np.uint32)
        assert np.array_equal(got[2], ref[2]) and got[3] == ref[3]
    none = hip.ransac_plane_samples(x, y, z, np.zeros((0, 3), np.uint32), 0.1)
    rnone = oracle.ransac_plane_samples(x, y, z, np.zeros((0, 3), np.uint32), 0.1)
    assert list(none[0]) == [0.0, 0.0, 1.0] and none[1] == 0.0 and np.array_equal(none[2], rnone[2])


def test_euclidean_cluster_parity(hip, oracle):
    """Clusters identical to the CPU restatement: blobs, a chain crossing many cells, duplicates,
    non-finite points (singletons), size filters, ties between equal-sized clusters."""
    rng = np.random.default_rng(3)
    blobs = [rng.normal(c, s, size=(m, 3)) for c, s, m in
             (((0, 0, 0), 0.3, 4000), ((10, 0, 0), 0.2, 2500), ((0, 12, 3), 0.5, 2500),
              ((-8, -8, 1), 0.1, 700))]
    chain = np.stack([np.arange(600) * 0.09 + 20.0, np.zeros(600), np.zeros(600)], 1)
    noise = rng.uniform(-30, 30, size=(3000, 3))
    pts = np.concatenate(blobs + [chain, noise]).astype(F)
    pts = pts[rng.permutation(len(pts))]
    dx, dy, dz = gen.degenerate_cloud(200, 4)
    x = np.concatenate([pts[:, 0], dx, np.array([np.nan, 5.0], F)])
    y = np.concatenate([pts[:, 1], dy, np.array([1.0, np.inf], F)])
    z = np.concatenate([pts[:, 2], dz, np.array([1.0, 2.0], F)])
    for thr, mn, mx in ((0.1, 1, 10 ** 9), (0.25, 5, 10 ** 9), (0.6, 2, 3000), (2.0, 1, 50), (0.1, 1, 1)):
        got = hip.euclidean_cluster(x, y, z, thr, mn, mx)
        ref = oracle.euclidean_cluster(x, y, z, thr, mn, mx)
        assert len(got) == len(ref), (thr, mn, mx)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), (thr, mn, mx)
    assert hip.euclidean_cluster(x, y, z, 0.0, 1, 10) == []
    assert hip.euclidean_cluster(x, y, z, 1.0, 0, 10) == []


def test_obstacle_pipeline_matches_oracle(hip, oracle):
    """The reference's obstacle-detection pipeline (examples/python/kitti_obstacle_detection.py:87-124:
    voxel 0.15 -> SOR k=20 x2.0 -> RANSAC ground 0.15 / 500 -> drop ground -> clusters 0.8 / 10 /
    20000) stage by stage against the CPU restatement, each stage fed with the previous stage's
    GPU output."""
    import pointclouds_rs as pc

    x, y, z = gen.kitti_scene(43, 40000, 2500, 800)
    g = hip.voxel_downsample(x, y, z, 0.15)
    r = oracle.voxel_downsample(x, y, z, 0.15)
    for a, b in zip(g, r):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    keep, n_keep, thr, _ = hip.sor(*g, 20, 2.0)
    rkeep, rn, rthr, _ = oracle.sor(*g, 20, 2.0, algo=1)
    assert np.array_equal(keep, rkeep) and n_keep == rn and np.float32(thr) == np.float32(rthr)
    c = [col[keep] for col in g]
    smp = pc.sample_triples(len(c[0]), 500, 1234)
    nrm, d, inl, best = hip.ransac_plane_samples(*c, smp, 0.15)
    rnrm, rd, rinl, rbest = oracle.ransac_plane_samples(*c, smp, 0.15)
    assert np.array_equal(nrm.view(np.uint32), rnrm.view(np.uint32)) and d == rd and best == rbest
    assert np.array_equal(inl, rinl) and abs(nrm[2]) > 0.95 and len(inl) > 0.5 * len(c[0])
    mask = np.ones(len(c[0]), dtype=bool)
    mask[inl] = False
    obst = [col[mask] for col in c]
    cl = hip.euclidean_cluster(*obst, 0.8, 10, 20000)
    rcl = oracle.euclidean_cluster(*obst, 0.8, 10, 20000)
    assert len(cl) == len(rcl) >= 3
    for a, b in zip(cl, rcl):
        assert np.array_equal(a, b)


def test_obstacle_pipeline_in_one_call_matches_the_stages(hip, oracle):
    """pcrs_obstacle_pipeline (one upload, intermediate clouds on the device) against the same
    stages run one by one on the oracle: counts, threshold, plane, obstacle cloud and clusters."""
    import pointclouds_rs as pc

    for seed, sizes in ((43, (40000, 2500, 800)), (7, (9000, 500, 100))):
        x, y, z = gen.kitti_scene(seed, *sizes)
        out = hip.obstacle_pipeline(x, y, z, 0.15, 20, 2.0, 0.15, 500, 1234, 0.8, 10, 20000)
        g = oracle.voxel_downsample(x, y, z, 0.15)
        rkeep, rn, rthr, _ = oracle.sor(*g, 20, 2.0, algo=1)
        c = [col[rkeep] for col in g]
        smp = pc.sample_triples(len(c[0]), 500, 1234)
        rnrm, rd, rinl, rbest = oracle.ransac_plane_samples(*c, smp, 0.15)
        mask = np.ones(len(c[0]), dtype=bool)
        mask[rinl] = False
        obst = [col[mask] for col in c]
        rcl = oracle.euclidean_cluster(*obst, 0.8, 10, 20000)
        assert out["n_voxels"] == len(g[0]) and out["n_kept"] == rn and out["n_ground"] == len(rinl)
        assert _bit_equal(out["sor_threshold"], rthr)
        assert _bit_equal(out["plane_normal"], rnrm) and out["plane_d"] == rd
        assert out["n_obstacles"] == len(obst[0])
        for a, b in zip(out["obstacles"], obst):
            assert _bit_equal(a, b)
        assert len(out["clusters"]) == len(rcl) >= 1
        for a, b in zip(out["clusters"], rcl):
            assert np.array_equal(a, b)
    # degenerate frames: nothing, one point, k = 0 (statistical_outlier.rs:5-12)
    e = np.empty(0, dtype=F)
    out = hip.obstacle_pipeline(e, e, e)
    assert out["n_voxels"] == 0 and out["clusters"] == []
    out = hip.obstacle_pipeline([1.0], [2.0], [3.0], cluster_min_size=1)
    assert out["n_voxels"] == 1 and out["n_kept"] == 1 and out["n_ground"] == 0 and out["n_obstacles"] == 1
    assert [list(c) for c in out["clusters"]] == [[0]]
    x, y, z = gen.kitti_scene(7, 3000, 100, 50)
    out = hip.obstacle_pipeline(x, y, z, sor_k=0)
    assert out["n_voxels"] > 0 and out["n_kept"] == 0 and out["clusters"] == []


def test_select_dev_is_pointcloud_select(hip):
    from pointclouds_rs.dist import HipEngine
    import ctypes as C

    eng = HipEngine.shared(0)
    rng = np.random.default_rng(4)
    n = 100_003
    cols = [rng.normal(size=n).astype(F) for _ in range(3)]
    mask = (rng.random(n) < 0.37).astype(np.uint8)
    d = [eng.from_numpy(c) for c in cols]
    dm = eng.from_numpy(mask)
    for invert in (0, 1):
        o = [eng.empty(n) for _ in range(3)]
        oi = eng.empty(n, "int32")
        cnt = C.c_size_t()
        eng.ctx.check(eng.lib.pcrs_select_dev(eng.ctx.handle, *[eng._p(c) for c in d], n, eng._p(dm), invert,
                                              *[eng._p(c) for c in o], eng._p(oi), C.byref(cnt)))
        want = np.flatnonzero((mask != 0) != bool(invert))
        assert cnt.value == len(want)
        assert np.array_equal(oi.cpu().numpy()[: len(want)].view(np.uint32), want.astype(np.uint32))
        for got, c in zip(o, cols):
            assert np.array_equal(got.cpu().numpy()[: len(want)], c[want])


def test_knn_strays_over_dense_surface(hip, oracle):
    """Stray points high above (and far beside) a dense plane: their neighbours lie many cells
    away, in a small patch of the plane.  The retry stage bounds their K-th distance from a sample
    of a large cube and then searches only the ball of that radius; identities, order and distances
    must still be those of the exhaustive search, as lists (k = 21, 8) and through the consumers."""
    rng = np.random.default_rng(77)
    n_plane, n_stray = 60000, 400
    plane = np.stack([rng.uniform(-20, 20, n_plane), rng.uniform(-20, 20, n_plane),
                      rng.uniform(-0.05, 0.05, n_plane)], 1)
    stray = np.stack([rng.uniform(-30, 30, n_stray), rng.uniform(-30, 30, n_stray),
                      rng.uniform(-6, 14, n_stray)], 1)
    far = np.array([[150.0, -90.0, 40.0], [-200.0, 10.0, 0.0], [0.0, 0.0, 300.0]])
    pts = np.concatenate([plane, stray, far]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    x, y, z = (np.ascontiguousarray(pts[:, a]) for a in range(3))
    sparse = np.nonzero(np.abs(z) > 0.06)[0]
    for k in (21, 8):
        ix = hip.Index(x, y, z, k_hint=k)
        idx, dist, cnt = ix.knn(x[sparse], y[sparse], z[sparse], k)
        st = ix.last_stats()
        assert st["grid_ring"] > 100 and st["brute"] == 0  # the strays do take the retry stage
        ridx, rdist, _, rcnt = oracle.knn(x, y, z, x[sparse], y[sparse], z[sparse], k, algo=1)
        assert np.array_equal(cnt, rcnt) and np.array_equal(idx, ridx)
        assert np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
        ix.close()
    keep, n_keep, thr, mean = hip.sor(x, y, z, 20, 2.0)
    rkeep, rn, rthr, rmean = oracle.sor(x, y, z, 20, 2.0, algo=1)
    assert np.array_equal(mean.view(np.uint32), rmean.view(np.uint32))
    assert np.array_equal(keep, rkeep) and n_keep == rn and np.float32(thr) == np.float32(rthr)
    nrm = hip.normals(x, y, z, 16)
    rnrm = oracle.normals(x, y, z, 16, algo=1)
    ok = normals_close(nrm, rnrm, 1e-5)
    assert ok.mean() > 0.9999 and ok[sparse].mean() > 0.99, (ok.mean(), ok[sparse].mean())


def test_apply_transform_bit_exact(hip, oracle):
    x, y, z = gen.uniform_cloud(10000, 4, -50, 50)
    R = gen.rot_z(0.3)
    t = (0.25, -3.0, 7.5)
    got = hip.apply_transform(x, y, z, R, t)
    ref = oracle.apply_transform(x, y, z, R, t)
    for a, b in zip(got, ref):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    r1, t1 = hip.compose(R, t, gen.rot_z(-0.1), (1, 2, 3))
    r2, t2 = oracle.compose(R, t, gen.rot_z(-0.1), (1, 2, 3))
    assert np.array_equal(r1, r2) and np.array_equal(t1, t2)


def _icp_pair(n, seed=99):
    tgt = gen.hemisphere(n, seed, 5.0)
    return tgt


@pytest.mark.parametrize("algo", ALGOS, ids=["brute", "grid"])
def test_correspondences_and_step_sums(hip, oracle, algo):
    from pointclouds_rs import _ffi

    tgt = _icp_pair(5000)
    tn = oracle.normals(*tgt, 16, algo=1)
    src = oracle.apply_transform(*tgt, gen.rot_z(0.05), (0.3, -0.2, 0.1))
    ix = hip.Index(*tgt, k_hint=1, algo=algo)
    for max_d in (np.inf, 0.25):
        t, d, n = ix.find_correspondences(*src, max_d)
        rt, rd, rn = oracle.find_correspondences(*src, *tgt, max_d, algo=1)
        assert n == rn and np.array_equal(t, rt)
        assert np.array_equal(d.view(np.uint32), rd.view(np.uint32))
    ix.close()
    icp = hip.Icp(src, tgt, tn)
    for max_d in (np.inf, 0.25):
        ref = oracle.icp_step(src, tgt, tn, max_d, algo=1)
        s = icp.accumulate(_ffi.ICP_POINT_TO_PLANE, max_d)
        assert s[0] == ref["n_corr"]
        assert np.isclose(s[1], ref["sum_d2"], rtol=1e-12)
        ata = ref["plane"][:36].reshape(6, 6)
        tri = [ata[r, c] for r in range(6) for c in range(r, 6)]
        assert np.allclose(s[2:23], tri, rtol=1e-11, atol=1e-12)
        assert np.allclose(s[23:29], ref["plane"][36:], rtol=1e-11, atol=1e-12)
        p = icp.accumulate(_ffi.ICP_POINT_TO_POINT, max_d)
        assert p[0] == ref["n_corr"]
        assert np.allclose(p[2:17], ref["point"], rtol=1e-11, atol=1e-12)
        # the f64 sum the accumulate call returns vs the reference's sequential f32 sum (which
        # pcrs_icp_run reproduces exactly): within 1e-5 relative
        assert abs(np.sqrt(s[1] / s[0]) - ref["rmse"]) <= 1e-5 * ref["rmse"]
    icp.close()


def test_knn_large_external_batch(hip, oracle):
    """Batches of 32768+ explicit queries are sorted by grid cell inside the library; results
    must still come back in the caller's order, defaults for non-finite queries included."""
    x, y, z = gen.uniform_cloud(50000, 12, 0, 30)
    rng = np.random.default_rng(4)
    q = rng.uniform(-5, 35, size=(40000, 3)).astype(F)
    q[::997, 1] = np.nan
    q[5::1999, 2] = np.inf
    for k in (1, 7):
        ix = hip.Index(x, y, z, k_hint=k, algo=2)
        idx, dist, cnt = ix.knn(q[:, 0], q[:, 1], q[:, 2], k)
        ix.close()
        ridx, rdist, _, rcnt = oracle.knn(x, y, z, q[:, 0], q[:, 1], q[:, 2], k, algo=1)
        assert np.array_equal(cnt, rcnt) and np.array_equal(idx, ridx)
        assert np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
    ix = hip.Index(x, y, z, k_hint=1, algo=2)
    t, d, n = ix.find_correspondences(q[:, 0], q[:, 1], q[:, 2], 2.0)
    ix.close()
    rt, rd, rn = oracle.find_correspondences(q[:, 0], q[:, 1], q[:, 2], x, y, z, 2.0, algo=1)
    assert n == rn and np.array_equal(t, rt) and np.array_equal(d.view(np.uint32), rd.view(np.uint32))


def test_correspondences_far_and_crowded_queries(hip, oracle):
    """1-NN queries that leave the 27-cell cube: far outside the cloud (escalating radii up to
    the whole grid), inside a crowded cell (handed to the wave kernel), non-finite."""
    tx, ty, tz = gen.uniform_cloud(30000, 3, 0, 10)
    blob = np.random.default_rng(5).normal(5.0, 1e-3, size=(6000, 3)).astype(F)
    tx, ty, tz = np.concatenate([tx, blob[:, 0]]), np.concatenate([ty, blob[:, 1]]), np.concatenate([tz, blob[:, 2]])
    q = np.concatenate([
        np.random.default_rng(6).uniform(-500, 500, size=(3000, 3)),
        np.random.default_rng(7).normal(5.0, 2e-3, size=(3000, 3)),
        np.random.default_rng(8).uniform(0, 10, size=(3000, 3)),
        np.array([[np.nan, 0, 0], [1e30, 1e30, -1e30], [5, 5, np.inf]]),
    ]).astype(F)
    ix = hip.Index(tx, ty, tz, k_hint=1, algo=2)
    for max_d in (np.inf, 3.0):
        t, d, n = ix.find_correspondences(q[:, 0], q[:, 1], q[:, 2], max_d)
        rt, rd, rn = oracle.find_correspondences(q[:, 0], q[:, 1], q[:, 2], tx, ty, tz, max_d, algo=1)
        assert n == rn and np.array_equal(t, rt)
        assert np.array_equal(d.view(np.uint32), rd.view(np.uint32))
    ix.close()


@pytest.mark.parametrize("n", [1000, 20000], ids=["unordered", "cell-ordered"])
def test_icp_current_keeps_caller_order(hip, oracle, n):
    """Large sources are reordered spatially inside the ICP object; the caller never sees it."""
    from pointclouds_rs import _ffi

    tgt = gen.hemisphere(n, 7, 5.0)
    src = oracle.apply_transform(*tgt, gen.rot_z(0.03), (0.1, -0.05, 0.02))
    rot, t = gen.rot_z(-0.01), (0.01, 0.02, -0.03)
    icp = hip.Icp(src, tgt, None)
    s0 = icp.accumulate(_ffi.ICP_POINT_TO_POINT, np.inf)
    icp.apply(rot, t)
    cur = icp.current()
    ref = oracle.apply_transform(*oracle.apply_transform(*src, np.eye(3, dtype=np.float32), (0, 0, 0)),
                                 rot, t)
    for a in range(3):
        assert np.array_equal(cur[a].view(np.uint32), ref[a].view(np.uint32))
    r = oracle.icp_step(src, tgt, None, np.inf, algo=1)
    assert s0[0] == r["n_corr"] and np.isclose(s0[1], r["sum_d2"], rtol=1e-12)
    icp.close()


def _bits(v):
    return np.asarray(v, dtype=F).view(np.uint32)


def _assert_same_run(res, ref):
    """The whole outcome of a registration: where the loop ended and why and the metrics bit
    for bit, the accumulated transform to a few units in the last place of 1.0 (on the noise
    floor an entry of 1e-9 of an increment may round the other way without moving any point)."""
    assert int(res.num_iterations) == int(ref.num_iterations)
    assert bool(res.converged) == bool(ref.converged)
    assert _bits(res.fitness) == _bits(ref.fitness)
    assert _bits(res.rmse) == _bits(ref.rmse), (res.rmse, ref.rmse)
    assert np.allclose(np.array(res.rotation), np.array(ref.rot), rtol=0, atol=2.5e-7)
    assert np.allclose(np.array(res.translation), np.array(ref.t), rtol=0, atol=2.5e-7)


# Which oracle loop the library must follow bit for bit: point-to-plane sums in f64 in the
# reference itself (icp_plane.rs:148-180); for point-to-point the library sums in f64 where the
# reference sums in f32 (icp.rs:221-244), so its twin is the oracle's f64 variant (mode 2) and
# the reference's own arithmetic (mode 0) is met within the noise of those f32 sums.
ORACLE_TWIN = {0: 2, 1: 1}


@pytest.mark.parametrize("mode", [0, 1], ids=["point", "plane"])
def test_icp_run_matches_oracle(hip, oracle, mode):
    tgt = _icp_pair(4000)
    tn = oracle.normals(*tgt, 16, algo=1)
    src = oracle.apply_transform(*tgt, gen.rot_z(0.05), (0.3, -0.2, 0.1))
    # tolerance 0 never fires (|d rmse| < 0 is false): all iterations run, as in the benchmark
    res = hip.icp_register(mode, src, tgt, tn, 30, 0.0, np.inf)
    rc, ref, _ = oracle.icp(ORACLE_TWIN[mode], src, tgt, tn, max_iter=30, tol=0.0, algo=1)
    assert rc == 0 and ref.num_iterations == 30 and not ref.converged
    _assert_same_run(res, ref)
    if mode == 0:  # against the reference's f32 sums: equal up to their noise
        rc, lit, _ = oracle.icp(0, src, tgt, tn, max_iter=30, tol=0.0, algo=1)
        assert res.num_iterations == lit.num_iterations and res.fitness == lit.fitness
        assert np.allclose(np.array(res.rotation), np.array(lit.rot), atol=2e-5)
        assert np.allclose(np.array(res.translation), np.array(lit.t), atol=5e-5)
    # and the registration actually recovers the inverse motion
    assert np.allclose(np.array(res.rotation).reshape(3, 3), gen.rot_z(-0.05), atol=2e-2)


@pytest.mark.parametrize("n,tol,max_d", [(3000, 1e-5, 1.0), (3000, 1e-3, np.inf), (12000, 1e-6, 0.5),
                                         (700, 1e-4, np.inf)])
def test_icp_loop_control_matches_oracle(hip, oracle, n, tol, max_d):
    """With a live tolerance the loop must stop in the iteration in which the reference's stops:
    the rmse of every iteration is the reference's left-to-right f32 sum in source order
    (icp.rs:273-282), so num_iterations, converged and the rmse are identical, not merely close.
    n = 12000 runs with the source reordered inside the library, n = 700 on the brute-force
    index."""
    tgt = _icp_pair(n)
    tn = oracle.normals(*tgt, 16, algo=1)
    src = oracle.apply_transform(*tgt, gen.rot_z(0.03), (0.1, 0.05, -0.05))
    for mode in (0, 1):
        res = hip.icp_register(mode, src, tgt, tn, 50, tol, max_d)
        rc, ref, _ = oracle.icp(ORACLE_TWIN[mode], src, tgt, tn, max_iter=50, tol=tol, max_d=max_d,
                                algo=1)
        assert rc == 0 and 1 < ref.num_iterations < 50 and ref.converged
        _assert_same_run(res, ref)


def test_icp_metrics_pass_and_lost_correspondences(hip, oracle):
    tgt = _icp_pair(3000)
    src = oracle.apply_transform(*tgt, gen.rot_z(0.03), (0.1, 0.05, -0.05))
    # max_iterations == 0: one metrics pass (icp.rs:190-197)
    res = hip.icp_register(0, src, tgt, None, 0, 1e-5, np.inf)
    rc, ref, _ = oracle.icp(0, src, tgt, None, max_iter=0, tol=1e-5, algo=1)
    assert res.num_iterations == 0
    _assert_same_run(res, ref)
    # a gate nothing passes: the loop ends in its first iteration without metrics (icp.rs:163-165)
    far = oracle.apply_transform(*tgt, gen.rot_z(0.0), (100.0, 0.0, 0.0))
    for tol in (0.0, 1e-5):
        res = hip.icp_register(0, far, tgt, None, 20, tol, 1.0)
        rc, ref, _ = oracle.icp(2, far, tgt, None, max_iter=20, tol=tol, max_d=1.0, algo=1)
        assert ref.num_iterations == 1 and not ref.converged and np.isinf(ref.rmse)
        _assert_same_run(res, ref)


def test_kitti_scene_sor(hip, oracle):
    """tests/real_world_pipeline.rs:106-113 on the raw 57k-point scene: removes some, keeps most."""
    x, y, z = gen.kitti_scene(42)
    keep, n_keep, thr, mean = hip.sor(x, y, z, 20, 2.0)
    assert len(x) // 2 < n_keep < len(x)
    rkeep, rn, rthr, rmean = oracle.sor(x, y, z, 20, 2.0, algo=1)
    assert np.array_equal(mean.view(np.uint32), rmean.view(np.uint32))
    assert np.array_equal(keep, rkeep)


def test_python_surface_end_to_end(hip):
    """tests/test_python.py:98-173 of the reference, through the pointclouds_rs mirror."""
    import pointclouds_rs as pc

    rng = np.random.default_rng(0)
    cluster = (rng.random((50, 3)) * 0.1).astype(F)
    data = np.vstack([cluster, np.array([[100.0, 100.0, 100.0]], dtype=F)])
    cloud = pc.PointCloud.from_numpy(data)
    assert pc.statistical_outlier_removal(cloud, 10, 1.0).len() <= cloud.len()
    assert pc.radius_outlier_removal(cloud, 0.5, 3).len() < cloud.len()
    xs = np.linspace(-2, 2, 10, dtype=F)
    xx, yy = np.meshgrid(xs, xs)
    zz = np.random.default_rng(123).normal(0, 1e-4, 100).astype(F)
    tdata = np.column_stack([xx.ravel(), yy.ravel(), zz]).astype(F)
    target = pc.estimate_normals(pc.PointCloud.from_numpy(tdata), 10)
    assert target.len() == 100 and target.normals_numpy().shape == (100, 3)
    sdata = tdata.copy()
    sdata[:, 2] += 0.3
    res = pc.icp_point_to_plane(pc.PointCloud.from_numpy(sdata), target)
    assert res.converged and res.rmse < 0.1
    sq = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], dtype=F)
    res = pc.icp_point_to_point(pc.PointCloud.from_numpy(sq),
                                pc.PointCloud.from_numpy(sq + np.array([0.1, 0, 0], dtype=F)))
    assert res.converged and res.rmse < 0.1
    moved = pc.apply_transform(pc.PointCloud.from_numpy(sq), np.eye(3), [1, 2, 3])
    assert np.array_equal(moved.to_numpy(), sq + np.array([1, 2, 3], dtype=F))


def test_query_subrange_equals_full(oracle):
    """Query sharding: a sub-range [q_begin, q_begin + q_count) of the indexed cloud (the
    multi-GPU path) must give exactly the rows of the full-range result."""
    from pointclouds_rs.dist import HipEngine

    eng = HipEngine.shared(0)
    x, y, z = gen.degenerate_cloud(300, 7)
    ux, uy, uz = gen.uniform_cloud(30000, 11)
    x, y, z = np.concatenate([x, ux]), np.concatenate([y, uy]), np.concatenate([z, uz])
    n = len(x)
    cols = [eng.from_numpy(c) for c in (x, y, z)]
    rmean = oracle.sor_mean_dist(x, y, z, 16, algo=1)
    rn = oracle.normals(x, y, z, 16, (1.0, 2.0, 3.0), algo=1)
    for algo in (1, 2):
        ix = eng.build_index(*cols, 17, algo)
        for b, c in ((0, n), (0, 1), (123, 7001), (n - 5000, 5000), (17, 0)):
            mean = eng.empty(c); nx = eng.empty(c); ny = eng.empty(c); nz = eng.empty(c)
            eng.sor_normals_range(ix, b, c, 16, 16, (1.0, 2.0, 3.0), mean, nx, ny, nz)
            eng.torch.cuda.synchronize()
            assert np.array_equal(mean.cpu().numpy().view(np.uint32), rmean[b:b + c].view(np.uint32))
            for got, ref in zip((nx, ny, nz), rn):
                g, r = got.cpu().numpy(), ref[b:b + c]
                assert np.abs(g.astype(np.float64) - r).max(initial=0.0) <= 1e-5
        eng.destroy_index(ix)


def test_sor_host_call_large_cloud(hip, oracle):
    """pcrs_sor with the means returned (device statistics and mask, means downloaded beside
    them) on mid-size clouds with non-finite points: mask, count, threshold bit for bit."""
    for n in (150_000, 260_000):
        x, y, z = gen.uniform_cloud(n, 21)
        x[::9973] = np.nan  # a few non-finite points: their means are not part of the sums
        keep, n_keep, thr, mean = hip.sor(x, y, z, 8, 1.5)
        rkeep, rn, rthr, rmean = oracle.sor(x, y, z, 8, 1.5, algo=1)
        assert np.array_equal(mean.view(np.uint32), rmean.view(np.uint32))
        assert np.array_equal(keep, rkeep) and n_keep == rn
        assert F(thr).view(np.uint32) == F(rthr).view(np.uint32)


def test_slot_runs_equal_full(oracle):
    """Ownership by space: the runs of whole cells that pcrs_index_slot_bounds hands out
    partition the cloud, and the answers of a run are exactly the single-call answers of the
    points it names (means bit for bit, normals within tolerance)."""
    from pointclouds_rs.dist import HipEngine

    eng = HipEngine.shared(0)
    torch = eng.torch
    dx, dy, dz = (np.nan_to_num(c, nan=1.5, posinf=2.5, neginf=0.5) for c in gen.degenerate_cloud(300, 7))
    kx, ky, kz = gen.kitti_scene(5, 20000, 3000, 500)
    ux, uy, uz = gen.uniform_cloud(30000, 11)
    x, y, z = (np.concatenate(c).astype(F) for c in ((dx, kx, ux), (dy, ky, uy), (dz, kz, uz)))
    n = len(x)
    cols = [eng.from_numpy(c) for c in (x, y, z)]
    rmean = oracle.sor_mean_dist(x, y, z, 16, algo=1)
    rn = oracle.normals(x, y, z, 16, (1.0, 2.0, 3.0), algo=1)
    ix = eng.build_index(*cols, 17, 2)
    for parts in (1, 2, 3, 8, 50):
        bounds = eng.slot_bounds(ix, parts)
        assert bounds[0] == 0 and bounds[-1] == n and bounds == sorted(bounds)
        seen = np.zeros(n, dtype=np.int64)
        by_index = eng.empty(n)
        for r in range(parts):
            b, c = bounds[r], bounds[r + 1] - bounds[r]
            point = eng.empty(c, "int32"); mean = eng.empty(c)
            nx = eng.empty(c); ny = eng.empty(c); nz = eng.empty(c)
            eng.sor_normals_slots(ix, b, c, 16, 16, (1.0, 2.0, 3.0), point, mean, nx, ny, nz)
            eng.scatter(mean, point, by_index)
            torch.cuda.synchronize()
            pt = point.cpu().numpy().view(np.uint32)
            seen[pt] += 1
            assert np.array_equal(mean.cpu().numpy().view(np.uint32), rmean[pt].view(np.uint32))
            for got, ref in zip((nx, ny, nz), rn):
                assert np.abs(got.cpu().numpy().astype(np.float64) - ref[pt]).max(initial=0.0) <= 1e-5
        assert (seen == 1).all()
        assert np.array_equal(by_index.cpu().numpy().view(np.uint32), rmean.view(np.uint32))
    # a second build may order the points inside a cell differently, never the cells
    ix2 = eng.build_index(*cols, 17, 2)
    assert eng.slot_bounds(ix2, 8) == eng.slot_bounds(ix, 8)
    eng.destroy_index(ix2)
    eng.destroy_index(ix)
    # no cell order: brute-force index, cloud with a non-finite point
    ixb = eng.build_index(*cols, 17, 1)
    assert eng.slot_bounds(ixb, 4) == [0] * 5
    eng.destroy_index(ixb)
    xn = x.copy(); xn[17] = np.nan
    ixn = eng.build_index(eng.from_numpy(xn), cols[1], cols[2], 17, 2)
    assert eng.slot_bounds(ixn, 4) == [0] * 5
    eng.destroy_index(ixn)


def test_sharded_driver_cells_single_rank(oracle):
    """pointclouds_rs.dist with ownership by cells on one rank: the whole pipeline (search over
    the run, scatter of the means to index order, exact threshold, mask) against the oracle."""
    from pointclouds_rs.dist import Collectives, HipEngine, sor_normals_sharded

    eng = HipEngine.shared(0)
    x, y, z = gen.kitti_scene(9, 30000, 4000, 800)
    n = len(x)
    shard = [eng.from_numpy(c) for c in (x, y, z)]
    keep, n_keep, thr, mean = oracle.sor(x, y, z, 16, 1.0, algo=1)
    rn = oracle.normals(x, y, z, 16, algo=1)
    res = sor_normals_sharded(eng, Collectives(None), shard, n, 16, 1.0, 16, ownership="cells")
    assert res["ownership"] == "cells"
    pt = res["point"].cpu().numpy().view(np.uint32)
    assert np.array_equal(np.sort(pt), np.arange(n, dtype=np.uint32))
    assert F(res["thr"]).view(np.uint32) == F(thr).view(np.uint32) and res["n_keep"] == n_keep
    assert np.array_equal(res["mean"].cpu().numpy().view(np.uint32), mean[pt].view(np.uint32))
    assert np.array_equal(res["keep"].cpu().numpy().astype(bool), keep[pt])
    for got, ref in zip((res["nx"], res["ny"], res["nz"]), rn):
        assert np.abs(got.cpu().numpy().astype(np.float64) - ref[pt]).max() <= 1e-5
    # a cloud without cell order falls back to ownership by index
    xs = [eng.from_numpy(c[:40]) for c in (x, y, z)]
    r2 = sor_normals_sharded(eng, Collectives(None), xs, 40, 8, 1.0, 8, ownership="cells")
    assert r2["ownership"] == "index" and r2["point"] is None


def test_large_cloud_parity(oracle):
    """6 M points (the benchmark's per-GPU size times six): means, threshold and mask bit-exact,
    normals within tolerance on a sample; also the one-level (global-atomic) sort, which grids
    beyond 33 M cells fall back to, must give the same answers."""
    import os

    from pointclouds_rs.dist import HipEngine

    eng = HipEngine.shared(0)
    n = 6_000_000
    side = np.float32(100.0 * 6 ** (1.0 / 3.0))
    cols_h = [(gen.uniform01(77, n, a) * side).astype(F) for a in range(3)]
    cols = [eng.from_numpy(c) for c in cols_h]
    rmean = oracle.sor_mean_dist(*cols_h, 16, algo=1, threads=0)
    ok_ref, thr_ref, _, _ = oracle.sor_threshold(rmean, 1.0)
    results = []
    for sort in ("two-level", "atomic"):
        if sort == "atomic":
            os.environ["PCRS_GRID_SORT"] = "atomic"
        try:
            ix = eng.build_index(*cols, 17)
        finally:
            os.environ.pop("PCRS_GRID_SORT", None)
        mean = eng.empty(n); nx = eng.empty(n); ny = eng.empty(n); nz = eng.empty(n)
        eng.sor_normals_range(ix, 0, n, 16, 16, (0.0, 0.0, 0.0), mean, nx, ny, nz)
        keep = eng.empty(n, "uint8")
        thr, ok, n_keep = eng.select(mean, 1.0, mean, keep)
        eng.destroy_index(ix)
        assert np.array_equal(mean.cpu().numpy().view(np.uint32), rmean.view(np.uint32)), sort
        assert ok == ok_ref and F(thr).view(np.uint32) == F(thr_ref).view(np.uint32)
        k = keep.cpu().numpy().astype(bool)
        assert np.array_equal(k, rmean <= F(thr_ref)) and n_keep == int(k.sum())
        results.append([t.cpu().numpy() for t in (nx, ny, nz)])
    for a, b in zip(*results):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # normals of a slice against the oracle (a whole-cloud oracle run would take minutes)
    sl = slice(1_000_000, 1_020_000)
    idx, _, _, _ = oracle.knn(*cols_h, cols_h[0][sl], cols_h[1][sl], cols_h[2][sl], 16, algo=1)
    assert idx.shape == (20000, 16)
    pts = np.stack(cols_h, 1)
    nb = pts[idx]                                   # (20000, 16, 3)
    c = nb.mean(1, dtype=np.float64)
    d = nb - c[:, None, :]
    cov = np.einsum("nki,nkj->nij", d, d)
    w, v = np.linalg.eigh(cov)
    ref = v[:, :, 0]
    got = np.stack([r[sl] for r in results[0]], 1).astype(np.float64)
    assert np.all(np.abs(np.abs((got * ref).sum(1)) - 1.0) < 1e-3)  # same axis (sign set by viewpoint)


def test_search_stage_statistics():
    """The retry stages must actually be exercised by clouds with sparse outliers."""
    from pointclouds_rs import hip

    x, y, z = gen.kitti_scene(42)
    ix = hip.Index(x, y, z, k_hint=17, algo=2)
    ix.knn(x, y, z, 17)
    st = ix.last_stats()
    ix.close()
    assert st["grid_tile"] + st["grid_ring"] + st["brute"] == len(x)
    assert st["grid_tile"] > 0.9 * len(x) and st["grid_ring"] > 0


def test_correspondences_bounded_on_a_target_beyond_the_whole_grid_scan(hip, oracle):
    """A target of more than 4 M points, queries inside, around and far outside it: a wave does
    not scan such a cloud point by point, it samples it and searches the ball of the distance the
    sample gives.  With a finite max_correspondence_distance the searches also end at the cube
    that covers that distance."""
    tx, ty, tz = gen.uniform_cloud(4_300_000, 21, 0, 100)
    rng = np.random.default_rng(9)
    q = np.concatenate([
        rng.uniform(0, 100, size=(3000, 3)),      # inside the cloud
        rng.uniform(-3, 103, size=(1500, 3)),     # around its faces
        rng.uniform(-400, 500, size=(150, 3)),    # far outside
        np.array([[np.nan, 1, 1], [50, 50, np.inf]]),
    ]).astype(F)
    ix = hip.Index(tx, ty, tz, k_hint=1, algo=2)
    for max_d in (0.4, 2.5, np.inf):
        t, d, n = ix.find_correspondences(q[:, 0], q[:, 1], q[:, 2], max_d)
        rt, rd, rn = oracle.find_correspondences(q[:, 0], q[:, 1], q[:, 2], tx, ty, tz, max_d, algo=1)
        assert n == rn and np.array_equal(t, rt)
        assert np.array_equal(d.view(np.uint32), rd.view(np.uint32))
        assert ix.last_stats()["brute"] == 0
    ix.close()


def test_knn_strays_around_a_large_cloud(hip, oracle):
    """Queries far from a cloud of more than 4 M points, k > 1: the retry stage bounds their K-th
    distance from samples (of the whole grid, then of the ball that bound gives, as often as the
    region is large) and searches only the last ball; they do not go to the brute-force stage."""
    tx, ty, tz = gen.uniform_cloud(4_300_000, 23, 0, 100)
    rng = np.random.default_rng(10)
    q = np.concatenate([
        rng.uniform(0, 100, size=(500, 3)),       # inside the cloud
        rng.uniform(-3, 103, size=(500, 3)),      # around its faces
        rng.uniform(-400, 500, size=(200, 3)),    # far outside
        np.array([[50.0, 50.0, 5000.0], [-1e4, -1e4, -1e4], [3e9, 0.0, -2e12], [1e30, 1e30, -1e30]]),
    ]).astype(F)
    ix = hip.Index(tx, ty, tz, k_hint=16, algo=2)
    for k in (16, 5):
        idx, dist, cnt = ix.knn(q[:, 0], q[:, 1], q[:, 2], k)
        st = ix.last_stats()
        # (the last two queries are so far away that every point of the cloud is at the same f32
        # distance from them, +inf for the last: more ties than the retry stage lists, the two
        # brute-force queries)
        assert st["grid_ring"] >= 200 and st["brute"] <= 2
        ridx, rdist, _, rcnt = oracle.knn(tx, ty, tz, q[:, 0], q[:, 1], q[:, 2], k, algo=1)
        assert np.array_equal(cnt, rcnt) and np.array_equal(idx, ridx)
        assert np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
    ix.close()


def _tie_and_crowd_cloud():
    """A cloud that takes every side road of the one-walk search kernel: a uniform background,
    a blob of 3000 points inside one grid cell (cubes of more candidates than the kernel's
    10-bit ordinals can name), a patch of an integer lattice (runs of equal distances, also across
    the K-th place), and mirrored pairs around lattice points (isolated pairs of equal distance)."""
    rng = np.random.default_rng(2024)
    bg = rng.uniform(0, 40, size=(30000, 3))
    blob = 20.0 + rng.uniform(0, 0.05, size=(3000, 3))
    g = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(12), indexing="ij"),
                 -1).reshape(-1, 3) * 0.5 + 2.0
    centres = rng.uniform(5, 35, size=(2000, 3)).round(2)
    off = rng.uniform(0.05, 0.3, size=(2000, 3)).round(3)
    pairs = np.concatenate([centres, centres + off, centres - off])
    p = np.concatenate([bg, blob, g, pairs]).astype(F)
    p = p[rng.permutation(len(p))]
    return tuple(np.ascontiguousarray(p[:, i]) for i in range(3))


@pytest.mark.parametrize("k", [2, 3, 4, 8, 12, 16, 17, 21, 24])
def test_one_walk_kernel_side_roads(hip, oracle, k):
    """Lists of 2 to 24 neighbours go through the kernel whose insertion chain carries truncated
    distance keys: pairs it has to re-order by the full keys, runs and crowded cubes it hands
    over must all end in the oracle's lists, bit for bit."""
    x, y, z = _tie_and_crowd_cloud()
    ix = hip.Index(x, y, z, k_hint=k, algo=2)
    idx, dist, cnt = ix.knn(x, y, z, k)
    st = ix.last_stats()
    ix.close()
    ridx, rdist, _, rcnt = oracle.knn(x, y, z, x, y, z, k, algo=1)
    assert np.array_equal(cnt, rcnt)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(dist.view(np.uint32), rdist.view(np.uint32))
    assert st["grid_tile"] + st["grid_ring"] + st["brute"] == len(x)


def test_one_walk_kernel_consumers(hip, oracle):
    """The same cloud through the fused consumers: means, threshold and mask bit-exact, normals
    to the bar of this file."""
    x, y, z = _tie_and_crowd_cloud()
    keep, n_keep, thr, nrm = hip.sor_normals(x, y, z, 16, 1.0, 16)
    rkeep, rn, rthr, rmean = oracle.sor(x, y, z, 16, 1.0, algo=1)
    assert np.array_equal(keep, rkeep) and n_keep == rn
    assert F(thr).view(np.uint32) == F(rthr).view(np.uint32)
    rnrm = oracle.normals(x, y, z, 16, algo=1)
    # (equal distances make the 16-neighbour set of a lattice point a matter of index order,
    # which both sides resolve alike: the covariance sums are then the same numbers)
    for a, b in zip(nrm, rnrm):
        assert float(np.abs(a - b).max()) <= 1e-5


def _bit_equal(a, b):
    return np.array_equal(np.asarray(a, dtype=F).view(np.uint32), np.asarray(b, dtype=F).view(np.uint32))


def test_cloud_of_one_repeated_point(hip, oracle):
    """Every point in one place (kiddo panics on such a cloud; here it must be neither wrong nor
    slow): a grid of one cell, the lane kernel's zero-distance shortcut for SOR and normals,
    the exact stages for lists.  The oracle's (d2, descending index) order defines the lists."""
    import time

    n = 12_000
    x, y, z = np.full(n, 1.5, F), np.full(n, -2.25, F), np.full(n, 3.0, F)
    vp = (0.0, 0.0, 10.0)
    keep, n_keep, thr, mean = hip.sor(x, y, z, 16, 1.0)
    rkeep, rn, rthr, rmean = oracle.sor(x, y, z, 16, 1.0, algo=0)
    assert _bit_equal(mean, rmean) and n_keep == rn == n and np.array_equal(keep, rkeep)
    assert _bit_equal(thr, rthr) and thr == 0.0
    got = hip.normals(x, y, z, 16, vp)
    ref = oracle.normals(x, y, z, 16, vp, algo=0)
    for g, r in zip(got, ref):
        assert _bit_equal(g, r)
    keep2, nk2, thr2, nrm2 = hip.sor_normals(x, y, z, 16, 1.0, 16, vp)
    assert nk2 == n and thr2 == 0.0 and all(_bit_equal(g, r) for g, r in zip(nrm2, ref))
    # lists: the k largest indices, all at distance zero
    ix = hip.Index(x, y, z, k_hint=5)
    idx, dist, cnt = ix.knn(x[:300], y[:300], z[:300], 5)
    ix.close()
    ridx, rdist, _, rcnt = oracle.knn(x, y, z, x[:300], y[:300], z[:300], 5, algo=0)
    assert np.array_equal(idx, ridx) and np.array_equal(cnt, rcnt) and _bit_equal(dist, rdist)
    assert list(idx[0]) == [n - 1, n - 2, n - 3, n - 4, n - 5]
    # the size that used to take 0.7 s (every query compared with every point): analytic answer
    n = 100_000
    x, y, z = np.full(n, 1.0, F), np.full(n, 2.0, F), np.full(n, 3.0, F)
    hip.sor(x, y, z, 16, 1.0)
    t0 = time.perf_counter()
    keep, n_keep, thr, mean = hip.sor(x, y, z, 16, 1.0)
    dt = time.perf_counter() - t0
    assert n_keep == n and thr == 0.0 and not mean.any()
    assert dt < 0.05, f"{dt * 1e3:.1f} ms"
    got = hip.normals(x, y, z, 16, vp)
    assert not got[0].any() and not got[1].any() and np.all(got[2] == 1.0)


def test_half_the_cloud_is_one_point(hip, oracle):
    """10 K uniform points and 10 K copies of one of them, shuffled: the copies' cell holds more
    points than a lane walks (they are answered from the start of their cell), their neighbours'
    cubes go to the wave-per-query kernel."""
    ux, uy, uz = gen.uniform_cloud(10_000, 3)
    x = np.concatenate([ux, np.full(10_000, ux[17], F)])
    y = np.concatenate([uy, np.full(10_000, uy[17], F)])
    z = np.concatenate([uz, np.full(10_000, uz[17], F)])
    p = np.random.default_rng(1).permutation(len(x))
    x, y, z = x[p], y[p], z[p]
    vp = (50.0, 50.0, 500.0)
    for k in (16, 4):
        keep, n_keep, thr, mean = hip.sor(x, y, z, k, 1.0)
        rkeep, rn, rthr, rmean = oracle.sor(x, y, z, k, 1.0, algo=1)
        assert _bit_equal(mean, rmean), k
        assert _bit_equal(thr, rthr) and n_keep == rn and np.array_equal(keep, rkeep)
        keep2, nk2, thr2, nrm = hip.sor_normals(x, y, z, k, 1.0, k, vp)
        ref = oracle.normals(x, y, z, k, vp, algo=1)
        assert nk2 == rn and _bit_equal(thr2, rthr)
        assert normals_close(nrm, ref, 1e-5).all()
    ix = hip.Index(x, y, z, k_hint=17)
    q = slice(0, 400)
    idx, dist, cnt = ix.knn(x[q], y[q], z[q], 17)
    ix.close()
    ridx, rdist, _, rcnt = oracle.knn(x, y, z, x[q], y[q], z[q], 17, algo=0)
    assert np.array_equal(idx, ridx) and np.array_equal(cnt, rcnt) and _bit_equal(dist, rdist)


@pytest.mark.parametrize("world", [3, 8])
def test_slab_indexes_partition_the_cloud_and_match_the_oracle(oracle, world):
    """What each rank of a run with ownership by slabs computes, rank after rank on one GPU: the
    cuts, an index over the slab plus a halo, the slab's queries.  The slabs must partition the
    cloud and every answer must be the oracle's for the whole cloud -- for the points whose
    neighbours lie across a cut (in the halo) and for those whose neighbourhood reaches beyond the
    halo (answered from the index over the whole cloud: forced here by a second run with a halo a
    hundred times too small)."""
    from pointclouds_rs.dist import HipEngine

    eng = HipEngine.shared(0)
    x, y, z = gen.kitti_scene(5, 40000, 3000, 600)   # a ground plane, objects, strays
    n, k = len(x), 16
    cols = [eng.from_numpy(c) for c in (x, y, z)]
    rkeep, rn, rthr, rmean = oracle.sor(x, y, z, k, 1.0, algo=1)
    rnrm = oracle.normals(x, y, z, k, (0.0, 0.0, 1.7), algo=1)
    axis, cuts, halo, n_finite = eng.slab_cuts(*cols, world, k + 1)
    assert n_finite == n and cuts[0] == -np.inf and cuts[-1] == np.inf
    assert all(a <= b for a, b in zip(cuts, cuts[1:]))
    inf = float("inf")
    for shrink in (1.0, 0.01):
        seen = np.zeros(n, dtype=np.int32)
        escalated = 0
        for r in range(world):
            lo, hi, qlo, qhi = [-inf] * 3, [inf] * 3, [-inf] * 3, [inf] * 3
            qlo[axis], qhi[axis] = cuts[r], cuts[r + 1]
            lo[axis] = cuts[r] - halo * shrink if r > 0 else -inf
            hi[axis] = cuts[r + 1] + halo * shrink if r + 1 < world else inf
            ix = eng.build_index_box(*cols, k + 1, lo, hi)
            nq = eng.select_box(ix, qlo, qhi)
            assert nq is not None and abs(nq - n / world) < 0.02 * n + 64
            if shrink == 1.0:  # the three calls in one give the same slab
                ix2, nq2, axis2 = eng.slab_index(*cols, k + 1, world, r)
                assert ix2 is not None and nq2 == nq and axis2 == axis
                p2 = eng.empty(nq2, "int32"); m2 = eng.empty(nq2)
                eng.sor_normals_selected(ix2, k, 0, (0.0, 0.0, 0.0), p2, m2, None, None, None)
                eng.destroy_index(ix2)
                pt2 = p2.cpu().numpy().view(np.uint32)
                assert np.array_equal(m2.cpu().numpy().view(np.uint32), rmean[pt2].view(np.uint32))
            point = eng.empty(nq, "int32"); mean = eng.empty(nq)
            nx = eng.empty(nq); ny = eng.empty(nq); nz = eng.empty(nq)
            escalated += eng.sor_normals_selected(ix, k, k, (0.0, 0.0, 1.7), point, mean, nx, ny, nz)
            eng.destroy_index(ix)
            pt = point.cpu().numpy().view(np.uint32)
            seen[pt] += 1
            assert np.array_equal(mean.cpu().numpy().view(np.uint32), rmean[pt].view(np.uint32)), (world, r, shrink)
            got = [t.cpu().numpy() for t in (nx, ny, nz)]
            assert normals_close(got, [c[pt] for c in rnrm], 1e-5).all()
        assert (seen == 1).all()
        if shrink < 1.0:
            assert escalated > 0  # the small halo does send queries to the whole-cloud index


@pytest.mark.parametrize("cloud", ["rings", "blob", "rings4"])
def test_clouds_of_very_uneven_density_use_finer_and_coarser_levels_and_match_the_oracle(hip, oracle, cloud):
    """A spinning-LiDAR shaped scan (density falling by a factor of hundreds from the sensor
    outwards) and a millimetre blob in a box 200 units wide: the index gives the crowded part
    finer grids of its own, every point is answered at the finest level that holds it, and what
    a level cannot certify goes to the next coarser one.  Means, threshold, mask and lists must
    be the oracle's; the normals within 1e-5."""
    if cloud == "rings":
        x, y, z = gen.ring_scan(1)          # 131 072 points
        vp = (0.0, 0.0, 1.73)
    elif cloud == "rings4":
        # four sweeps: enough points far out, in cubes that cannot hold their neighbours, for the
        # index to add a grid of wider cells behind its own
        x, y, z = gen.ring_scan(4)          # 524 288 points
        vp = (0.0, 0.0, 1.73)
    else:
        x, y, z = gen.blob_in_box(30_000, 50_000)
        vp = (0.0, 0.0, 300.0)
    k = 16
    keep, n_keep, thr, mean = hip.sor(x, y, z, k, 1.0)
    rkeep, rn, rthr, rmean = oracle.sor(x, y, z, k, 1.0, algo=1)
    bad = np.flatnonzero(mean.view(np.uint32) != rmean.view(np.uint32))
    assert len(bad) == 0, (len(bad), bad[:5], mean[bad[:5]], rmean[bad[:5]])
    assert _bit_equal(thr, rthr) and n_keep == rn and np.array_equal(keep, rkeep)
    keep2, nk2, thr2, nrm = hip.sor_normals(x, y, z, k, 1.0, k, vp)
    ref = oracle.normals(x, y, z, k, vp, algo=1)
    assert nk2 == rn and _bit_equal(thr2, rthr) and np.array_equal(keep2, rkeep)
    assert normals_close(nrm, ref, 1e-5).mean() > 0.9999
    got = hip.normals(x, y, z, 5, vp)
    ref5 = oracle.normals(x, y, z, 5, vp, algo=1)
    assert normals_close(got, ref5, 1e-5).mean() > 0.999
    # explicit queries are answered by the coarsest grid alone (crowded cubes by the wave kernel)
    ix = hip.Index(x, y, z, k_hint=k + 1)
    finer, coarser = ix.levels()
    assert finer >= 1 and (coarser >= 1 or cloud != "rings4"), (finer, coarser)
    q = np.random.default_rng(2).choice(len(x), 3000, replace=False)
    idx, dist, cnt = ix.knn(x[q], y[q], z[q], k + 1)
    ix.close()
    ridx, rdist, _, rcnt = oracle.knn(x, y, z, x[q], y[q], z[q], k + 1, algo=1)
    assert np.array_equal(cnt, rcnt) and np.array_equal(idx, ridx) and _bit_equal(dist, rdist)


def test_several_crowded_regions_get_grids_of_their_own_and_match_the_oracle(hip, oracle):
    """Four dense clusters far apart in a sparse field: one box around the crowded cells would be
    most of the grid, so every cluster gets a finer grid of its own (side by side under the
    index's grid, searched on streams of their own beside the index's own queries) and hands the
    queries of its rim back.  Means, threshold, mask and lists must be the oracle's; the normals
    within 1e-5."""
    x, y, z = gen.clusters_in_field()   # 310 000 points
    vp = (0.0, 0.0, 300.0)
    k = 16
    ix = hip.Index(x, y, z, k_hint=k + 1)
    finer, coarser = ix.levels()
    assert finer >= 4, (finer, coarser)
    q = np.random.default_rng(3).choice(len(x), 3000, replace=False)
    idx, dist, cnt = ix.knn(x[q], y[q], z[q], k + 1)
    ix.close()
    ridx, rdist, _, rcnt = oracle.knn(x, y, z, x[q], y[q], z[q], k + 1, algo=1)
    assert np.array_equal(cnt, rcnt) and np.array_equal(idx, ridx) and _bit_equal(dist, rdist)
    rkeep, rn, rthr, rmean = oracle.sor(x, y, z, k, 1.0, algo=1)
    ref = oracle.normals(x, y, z, k, vp, algo=1)
    for _ in range(2):  # the second call finds the streams and the levels' buffers in place
        keep, n_keep, thr, mean = hip.sor(x, y, z, k, 1.0)
        bad = np.flatnonzero(mean.view(np.uint32) != rmean.view(np.uint32))
        assert len(bad) == 0, (len(bad), bad[:5], mean[bad[:5]], rmean[bad[:5]])
        assert _bit_equal(thr, rthr) and n_keep == rn and np.array_equal(keep, rkeep)
        keep2, nk2, thr2, nrm = hip.sor_normals(x, y, z, k, 1.0, k, vp)
        assert nk2 == rn and _bit_equal(thr2, rthr) and np.array_equal(keep2, rkeep)
        assert normals_close(nrm, ref, 1e-5).mean() > 0.9999


def test_aerial_scene_matches_the_oracle(hip, oracle):
    """A tenth of the reference's aerial example scene (examples/python/aerial_lidar.py:26-137:
    terrain surface, roofs and walls, canopies and trunks; 241 K points): SOR and normals against
    the oracle over the whole scene.  The scene's walls are exact planes x = const or y = const:
    the covariance of a neighbourhood on one has a row of exact zeros, and estimate.rs:139-238
    then takes its eigenvector from the cross product of a zero row and another -- rounding
    noise of the size of the last bit of cos(acos(..) / 3), which differs between two maths
    libraries.  Both sides return an in-plane vector for most wall points (the reference's
    algorithm is ill-conditioned there), and not for the same ones; every other normal of the
    scene must agree to 1e-5."""
    x, y, z, parts = gen.aerial_scene(0.1, with_parts=True)
    vp = (250.0, 250.0, 1000.0)
    k = 16
    keep, n_keep, thr, mean = hip.sor(x, y, z, k, 1.0)
    rkeep, rn, rthr, rmean = oracle.sor(x, y, z, k, 1.0, algo=1)
    assert np.array_equal(mean.view(np.uint32), rmean.view(np.uint32))
    assert _bit_equal(thr, rthr) and n_keep == rn and np.array_equal(keep, rkeep)
    keep2, nk2, thr2, nrm = hip.sor_normals(x, y, z, k, 1.0, k, vp)
    ref = oracle.normals(x, y, z, k, vp, algo=1)
    assert nk2 == rn and _bit_equal(thr2, rthr) and np.array_equal(keep2, rkeep)
    close = normals_close(nrm, ref, 1e-5)
    wall = np.zeros(len(x), dtype=bool)
    for name, b, e in parts:
        if name == "wall":
            wall[b:e] = True
    assert close[~wall].mean() > 0.99995, close[~wall].mean()
    assert close[wall].mean() > 0.95  # and most of the walls' (arbitrary) vectors too
