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


# ---- two ranks, two processes: on two GPUs where the machine has them, else both on the one GPU
# (which RCCL may refuse: the test is then skipped)

def _cloud2():
    from pointclouds_rs import synth

    return synth.kitti_scene(3, 51001, 2000, 1000)  # 58001 points: uneven shards


def _rank(rank, world, q_id, q_out, icp_inputs):
    import os
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(os.path.dirname(here), "pointclouds-rs_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch

    from pointclouds_rs import _ffi
    from pointclouds_rs.dist import HipEngine, shard_range

    # a GPU of its own where the machine has one for every rank (then RCCL really runs between
    # devices); on a one-GPU box both ranks take device 0, which RCCL may refuse
    eng = HipEngine.shared(rank if torch.cuda.device_count() >= world else 0)
    if rank == 0:
        uid = eng.dist_unique_id()
        for _ in range(world - 1):
            q_id.put(uid)
    else:
        uid = q_id.get(timeout=60)
    try:
        comm = eng.dist_comm_init(uid, rank, world)
    except _ffi.PcrsError as ex:
        q_out.put(dict(rank=rank, refused=str(ex)))
        return
    x, y, z = _cloud2()
    n = len(x)
    b, e = shard_range(n, world, rank)
    shard = [eng.from_numpy(c[b:e]) for c in (x, y, z)]
    r = eng.dist_sor_normals(comm, rank, world, shard, n, 16, 1.0, 16)
    torch.cuda.synchronize()
    out = dict(rank=rank, ownership=r["ownership"], thr=r["thr"], n_keep=r["n_keep"],
               point=r["point"].cpu().numpy().view(np.uint32).copy(), mean=r["mean"].cpu().numpy().copy(),
               keep=r["keep"].cpu().numpy().copy(),
               normals=[r[c].cpu().numpy().copy() for c in ("nx", "ny", "nz")])
    inp = np.load(icp_inputs)
    S = [inp["s%d" % a] for a in range(3)]
    sb, se = shard_range(len(S[0]), world, rank)
    d_src = [eng.from_numpy(np.ascontiguousarray(c[sb:se])) for c in S]
    d_tgt = [eng.from_numpy(inp["t%d" % a]) for a in range(3)]
    d_tn = [eng.from_numpy(inp["n%d" % a]) for a in range(3)]
    h = eng.icp_create(d_src, d_tgt, d_tn)
    try:
        for name, max_iter, tol in (("icp", 12, 0.0), ("icp_tol", 30, 1e-4)):
            res = eng.dist_icp_run(h, comm, rank, world, len(S[0]), 1, max_iter, tol, float("inf"))
            out[name] = dict(rot=list(res.rotation), t=list(res.translation), rmse=float(res.rmse),
                             fitness=float(res.fitness), iters=int(res.num_iterations),
                             converged=bool(res.converged))
    finally:
        eng.icp_destroy(h)
    q_out.put(out)
    eng.dist_comm_destroy(comm)


@pytest.mark.timeout(300)
def test_two_ranks_exchange_like_the_oracle_says(oracle, tmp_path):
    """The exchange between ranks (routing of the means, gathers, the ICP sums) with two
    processes.  A RCCL that refuses two ranks on one GPU skips the test on a one-GPU box."""
    import queue

    import torch.multiprocessing as mp
    from pointclouds_rs import synth

    world = 2
    tgt = synth.hemisphere(4000, 99, 5.0)
    tn = oracle.normals(*tgt, 12, algo=1)
    src = oracle.apply_transform(*tgt, synth.rot_z(0.04), (0.2, -0.1, 0.05))
    icp_inputs = str(tmp_path / "icp_inputs.npz")
    np.savez(icp_inputs, **{"s%d" % a: src[a] for a in range(3)},
             **{"t%d" % a: tgt[a] for a in range(3)}, **{"n%d" % a: tn[a] for a in range(3)})
    ctx = mp.get_context("spawn")
    q_id, q_out = ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, world, q_id, q_out, icp_inputs)) for r in range(world)]
    for p in procs:
        p.start()
    outs = []
    try:
        for _ in range(200):
            try:
                outs.append(q_out.get(timeout=1.0))
            except queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs):
                    break
            if len(outs) == world:
                break
    finally:
        for p in procs:
            p.join(20)
            if p.is_alive():
                p.kill()
    if any("refused" in o for o in outs):
        pytest.skip("RCCL does not put two ranks on one device: " + str([o.get("refused") for o in outs]))
    assert len(outs) == world, [p.exitcode for p in procs]
    outs.sort(key=lambda d: d["rank"])
    x, y, z = _cloud2()
    keep, n_keep, thr, mean = oracle.sor(x, y, z, 16, 1.0, algo=1)
    rn = oracle.normals(x, y, z, 16, algo=1)
    assert sum(o["n_keep"] for o in outs) == n_keep
    for o in outs:
        pt = o["point"]
        assert o["ownership"] == "slabs"
        assert F(o["thr"]).view(np.uint32) == F(thr).view(np.uint32)
        assert np.array_equal(o["mean"].view(np.uint32), mean[pt].view(np.uint32))
        assert np.array_equal(o["keep"].astype(bool), keep[pt])
        for got, ref in zip(o["normals"], rn):
            assert np.abs(got.astype(np.float64) - ref[pt]).max() <= 1e-5
    assert np.array_equal(np.sort(np.concatenate([o["point"] for o in outs])), np.arange(len(x)))
    for name, max_iter, tol in (("icp", 12, 0.0), ("icp_tol", 30, 1e-4)):
        rc, ref, _ = oracle.icp(1, src, tgt, tn, max_iter=max_iter, tol=tol, algo=1)
        assert outs[0][name] == outs[1][name]  # bit for bit the same on both ranks
        got = outs[0][name]
        assert got["iters"] == ref.num_iterations and got["converged"] == bool(ref.converged)
        assert got["fitness"] == ref.fitness and got["rmse"] == ref.rmse
        assert np.allclose(got["rot"], list(ref.rot), rtol=0, atol=2.5e-7)
        assert np.allclose(got["t"], list(ref.t), rtol=0, atol=2.5e-7)
