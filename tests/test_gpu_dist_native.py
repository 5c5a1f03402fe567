"""The library's own multi-GPU driver (pcrs_dist_*: collectives issued on RCCL inside the call)
with a communicator of ONE rank -- what a one-GPU box can run: the whole road (gather of the
columns, slab index, routing of the means to their index ranges, threshold) must give bit for bit
what the single-GPU entry points give.  The exchange between several ranks is the same code with
more peers; its logic is covered by the gloo tests of pointclouds_rs.dist."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F = np.float32


@pytest.fixture(scope="module")
def eng_comm():
    from pointclouds_rs.dist import HipEngine

    eng = HipEngine.shared(0)
    comm = eng.dist_comm_init(eng.dist_unique_id(), 0, 1)
    yield eng, comm
    eng.dist_comm_destroy(comm)


def _host(t):
    return None if t is None else t.cpu().numpy()


def _check_against_single_gpu(res, cloud, k_sor, std_mul, k_nrm):
    from pointclouds_rs import hip

    x, y, z = cloud
    n = len(x)
    point = _host(res["point"]).view(np.uint32)
    assert np.array_equal(np.sort(point), np.arange(n, dtype=np.uint32))  # one rank answers them all
    keep, n_keep, thr, mean = hip.sor(x, y, z, k_sor, std_mul)
    _, _, _, nrm = hip.sor_normals(x, y, z, k_sor, std_mul, k_nrm)
    assert res["n_keep"] == n_keep
    assert F(res["thr"]).view(np.uint32) == F(thr).view(np.uint32)
    assert np.array_equal(_host(res["mean"]).view(np.uint32), mean[point].view(np.uint32))
    assert np.array_equal(_host(res["keep"]).astype(bool), keep[point])
    for got, ref in zip((res["nx"], res["ny"], res["nz"]), nrm):
        assert np.array_equal(_host(got).view(np.uint32), ref[point].view(np.uint32))


@pytest.mark.timeout(120)
def test_one_rank_communicator_equals_the_single_gpu_call(eng_comm):
    from pointclouds_rs import synth

    eng, comm = eng_comm
    cloud = synth.kitti_scene(5, 51001, 2000, 1000)
    shard = [eng.from_numpy(c) for c in cloud]
    res = eng.dist_sor_normals(comm, 0, 1, shard, len(cloud[0]), 16, 1.0, 16)
    assert res["ownership"] == "slabs"
    _check_against_single_gpu(res, cloud, 16, 1.0, 16)
    # a second call on the same communicator, a smaller cloud, other list lengths
    cloud = synth.kitti_scene(6, 20011, 500, 300)
    shard = [eng.from_numpy(c) for c in cloud]
    res = eng.dist_sor_normals(comm, 0, 1, shard, len(cloud[0]), 8, 2.0, 12)
    _check_against_single_gpu(res, cloud, 8, 2.0, 12)


@pytest.mark.timeout(120)
def test_a_cloud_without_slabs_is_answered_by_index_ranges(eng_comm):
    from pointclouds_rs import synth

    eng, comm = eng_comm
    x, y, z = synth.kitti_scene(7, 30001, 1000, 500)
    x = x.copy()
    x[[5, 777, 20000]] = np.nan  # points without a place in space: no slabs to cut
    shard = [eng.from_numpy(c) for c in (x, y, z)]
    res = eng.dist_sor_normals(comm, 0, 1, shard, len(x), 16, 1.0, 16)
    assert res["ownership"] == "index"
    assert np.array_equal(_host(res["point"]).view(np.uint32), np.arange(len(x), dtype=np.uint32))
    _check_against_single_gpu(res, (x, y, z), 16, 1.0, 16)


@pytest.mark.timeout(120)
def test_normals_alone_and_argument_errors(eng_comm):
    from pointclouds_rs import _ffi, hip, synth

    eng, comm = eng_comm
    cloud = synth.kitti_scene(8, 20001, 500, 300)
    shard = [eng.from_numpy(c) for c in cloud]
    res = eng.dist_sor_normals(comm, 0, 1, shard, len(cloud[0]), 0, 1.0, 10)
    assert res["mean"] is None and res["keep"] is None and np.isnan(res["thr"])
    point = _host(res["point"]).view(np.uint32)
    ref = hip.normals(*cloud, 10)
    for got, want in zip((res["nx"], res["ny"], res["nz"]), ref):
        assert np.array_equal(_host(got).view(np.uint32), want[point].view(np.uint32))
    with pytest.raises(_ffi.PcrsError):  # the shard is not the rank's index range
        eng.dist_sor_normals(comm, 0, 1, shard, len(cloud[0]) + 1, 16, 1.0, 16)
    with pytest.raises(_ffi.PcrsError):  # outputs too small for the points this rank answers
        eng.dist_sor_normals(comm, 0, 1, shard, len(cloud[0]), 16, 1.0, 16, capacity=100)
    with pytest.raises(_ffi.PcrsError):  # rank outside the communicator
        eng.dist_sor_normals(comm, 1, 1, shard, len(cloud[0]), 16, 1.0, 16)
    # an empty cloud is nothing to do
    empty = [eng.empty(0) for _ in range(3)]
    res = eng.dist_sor_normals(comm, 0, 1, empty, 0, 16, 1.0, 16, capacity=1)
    assert res["n_keep"] == 0 and res["point"].numel() == 0


@pytest.mark.timeout(120)
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("max_iter,tol", [(12, 0.0), (30, 1e-4), (0, 0.0)])
def test_icp_over_a_one_rank_communicator_equals_icp_run(eng_comm, oracle, mode, max_iter, tol):
    from pointclouds_rs import hip, synth

    eng, comm = eng_comm
    tgt = synth.hemisphere(4000, 99, 5.0)
    tn = hip.normals(*tgt, 12)
    src = hip.apply_transform(*tgt, synth.rot_z(0.04), (0.2, -0.1, 0.05))
    single = hip.Icp(src, tgt, tn)
    try:
        ref = single.run(mode, max_iter, tol)
    finally:
        single.close()
    # the handle keeps the device columns it is given: they must outlive it
    d_src, d_tgt, d_tn = ([eng.from_numpy(c) for c in cols] for cols in (src, tgt, tn))
    h = eng.icp_create(d_src, d_tgt, d_tn)
    try:
        got = eng.dist_icp_run(h, comm, 0, 1, len(src[0]), mode, max_iter, tol, float("inf"))
    finally:
        eng.icp_destroy(h)
    assert got.num_iterations == ref.num_iterations and got.converged == ref.converged
    assert got.fitness == ref.fitness
    assert got.rmse == ref.rmse
    # one shard: the sums are the single-GPU sums, and so is every transform of the loop
    assert list(got.rotation) == list(ref.rotation)
    assert list(got.translation) == list(ref.translation)


@pytest.mark.timeout(60)
def test_icp_argument_checks(eng_comm):
    from pointclouds_rs import _ffi, synth

    eng, comm = eng_comm
    tgt = synth.hemisphere(500, 3, 5.0)
    d_tgt = [eng.from_numpy(c) for c in tgt]
    h = eng.icp_create(d_tgt, d_tgt, None)
    try:
        with pytest.raises(_ffi.PcrsError) as e:  # point-to-plane without normals (icp_plane.rs:27-32)
            eng.dist_icp_run(h, comm, 0, 1, 500, 1, 5, 0.0, float("inf"))
        assert e.value.code == _ffi.PCRS_ERR_NORMALS_MISMATCH
        with pytest.raises(_ffi.PcrsError):  # the shard is not the rank's range of the total
            eng.dist_icp_run(h, comm, 0, 1, 501, 0, 5, 0.0, float("inf"))
    finally:
        eng.icp_destroy(h)
