"""gloo tests (world size 2 and 3, CPU engine) of the query-sharded SOR + normals and the sharded
ICP (pointclouds_rs.dist): shard ranges, variable-size all-gathers, the f64 sum exchange and the
redundant solve must give every rank exactly what a single process computes."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    for p in (os.path.join(ROOT, "pointclouds-rs_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    import oracle_ffi
    from cpu_engine import CpuEngine
    from pointclouds_rs import _ffi, synth
    from pointclouds_rs.dist import Collectives, icp_sharded, shard_range, sor_normals_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = oracle_ffi.load()
    eng = CpuEngine(o)
    coll = Collectives(dist)
    x, y, z = synth.uniform_cloud(n_total, 42, 0, 20)
    b, e = shard_range(n_total, world, rank)
    shard = [eng.from_numpy(c[b:e]) for c in (x, y, z)]
    res = sor_normals_sharded(eng, coll, shard, n_total, 8, 1.0, 8)
    out = dict(rank=rank, begin=b, end=e, thr=res["thr"], n_keep=res["n_keep"],
               keep=res["keep"].numpy().copy(), mean=res["mean"].numpy().copy(),
               nx=res["nx"].numpy().copy())
    # the same cloud with ownership by runs of cells: results come with their point indices
    rc = sor_normals_sharded(eng, coll, shard, n_total, 8, 1.0, 8, ownership="cells")
    out["cells"] = dict(ownership=rc["ownership"], thr=rc["thr"], n_keep=rc["n_keep"],
                        point=rc["point"].numpy().view(np.uint32).copy(),
                        keep=rc["keep"].numpy().copy(), mean=rc["mean"].numpy().copy(),
                        nx=rc["nx"].numpy().copy())
    # ownership by slabs of space (each rank indexes its slab plus a halo only)
    rs = sor_normals_sharded(eng, coll, shard, n_total, 8, 1.0, 8, ownership="slabs")
    out["slabs"] = dict(ownership=rs["ownership"], thr=rs["thr"], n_keep=rs["n_keep"],
                        point=rs["point"].numpy().view(np.uint32).copy(),
                        keep=rs["keep"].numpy().copy(), mean=rs["mean"].numpy().copy(),
                        nx=rs["nx"].numpy().copy())
    # a cloud with a non-finite point has no slabs: every rank falls back to index ranges
    xb = x.copy()
    xb[5] = np.nan
    rb = sor_normals_sharded(eng, coll, [eng.from_numpy(c[b:e]) for c in (xb, y, z)], n_total, 8, 1.0,
                             0, ownership="slabs")
    out["slabs_fallback"] = rb["ownership"]
    # sharded ICP (point-to-plane), source split over the ranks, target replicated
    tgt = synth.hemisphere(1500, 99, 5.0)
    tn = o.normals(*tgt, 12, algo=1)
    src = o.apply_transform(*tgt, synth.rot_z(0.04), (0.2, -0.1, 0.05))
    sb, se = shard_range(len(src[0]), world, rank)
    r = icp_sharded(eng, coll, _ffi.ICP_POINT_TO_PLANE, [eng.from_numpy(c[sb:se]) for c in src],
                    len(src[0]), [eng.from_numpy(c) for c in tgt], [eng.from_numpy(c) for c in tn],
                    12, 0.0)
    out["icp"] = dict(rot=list(r["rotation"]), t=list(r["translation"]), rmse=r["rmse"],
                      fitness=r["fitness"], iters=r["num_iterations"])
    # the same with a live tolerance: the loop must end in the reference's iteration
    r = icp_sharded(eng, coll, _ffi.ICP_POINT_TO_PLANE, [eng.from_numpy(c[sb:se]) for c in src],
                    len(src[0]), [eng.from_numpy(c) for c in tgt], [eng.from_numpy(c) for c in tn],
                    30, 1e-4)
    out["icp_tol"] = dict(rot=list(r["rotation"]), t=list(r["translation"]), rmse=r["rmse"],
                          fitness=r["fitness"], iters=r["num_iterations"],
                          converged=r["converged"])
    q.put(out)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_matches_single_process(oracle):
    from pointclouds_rs import _ffi, synth
    from pointclouds_rs.dist import shard_range

    world, n_total = 2, 3001  # odd total: uneven shards
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in procs], key=lambda d: d["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    x, y, z = synth.uniform_cloud(n_total, 42, 0, 20)
    keep, n_keep, thr, mean = oracle.sor(x, y, z, 8, 1.0, algo=1)
    nx, _, _ = oracle.normals(x, y, z, 8, algo=1)
    assert [shard_range(n_total, world, r) for r in range(world)] == [(0, 1501), (1501, 3001)]
    assert sum(o["n_keep"] for o in outs) == n_keep
    for o in outs:
        b, e = o["begin"], o["end"]
        assert np.float32(o["thr"]) == np.float32(thr)  # every rank derives the same threshold
        assert np.array_equal(o["mean"], mean[b:e])
        assert np.array_equal(o["keep"].astype(bool), keep[b:e])
        assert np.array_equal(o["nx"], nx[b:e])
    # ownership by cells: the runs of the ranks partition the cloud, every answer is the
    # single-process answer of the point it names
    pts = np.concatenate([o["cells"]["point"] for o in outs])
    assert np.array_equal(np.sort(pts), np.arange(n_total, dtype=np.uint32))
    assert sum(o["cells"]["n_keep"] for o in outs) == n_keep
    for own in ("cells", "slabs"):
        pts = np.concatenate([o[own]["point"] for o in outs])
        assert np.array_equal(np.sort(pts), np.arange(n_total, dtype=np.uint32))
        assert sum(o[own]["n_keep"] for o in outs) == n_keep
        for o in outs:
            c = o[own]
            assert c["ownership"] == own and len(c["point"]) > 0
            assert np.float32(c["thr"]) == np.float32(thr)
            assert np.array_equal(c["mean"], mean[c["point"]])
            assert np.array_equal(c["keep"].astype(bool), keep[c["point"]])
            assert np.array_equal(c["nx"], nx[c["point"]])
    assert all(o["slabs_fallback"] == "index" for o in outs)
    # ICP: both ranks hold the same transform, equal to the single-process oracle run
    tgt = synth.hemisphere(1500, 99, 5.0)
    tn = oracle.normals(*tgt, 12, algo=1)
    src = oracle.apply_transform(*tgt, synth.rot_z(0.04), (0.2, -0.1, 0.05))
    rc, ref, _ = oracle.icp(1, src, tgt, tn, max_iter=12, tol=0.0, algo=1)
    assert outs[0]["icp"]["rot"] == outs[1]["icp"]["rot"] and outs[0]["icp"]["t"] == outs[1]["icp"]["t"]
    for o in outs:
        assert o["icp"]["iters"] == ref.num_iterations == 12
        assert o["icp"]["fitness"] == ref.fitness
        assert abs(o["icp"]["rmse"] - ref.rmse) <= 1e-4 * ref.rmse + 1e-6
        assert np.allclose(o["icp"]["rot"], list(ref.rot), atol=2e-5)
        assert np.allclose(o["icp"]["t"], list(ref.t), atol=5e-5)
    rc, ref, _ = oracle.icp(1, src, tgt, tn, max_iter=30, tol=1e-4, algo=1)
    assert 1 < ref.num_iterations < 30 and ref.converged
    assert outs[0]["icp_tol"] == outs[1]["icp_tol"]  # bit for bit the same on both ranks
    for o in outs:
        assert o["icp_tol"]["iters"] == ref.num_iterations
        assert o["icp_tol"]["converged"] == bool(ref.converged)
        assert o["icp_tol"]["fitness"] == ref.fitness
        assert abs(o["icp_tol"]["rmse"] - ref.rmse) <= 1e-4 * ref.rmse + 1e-6
        assert np.allclose(o["icp_tol"]["rot"], list(ref.rot), atol=2e-5)
        assert np.allclose(o["icp_tol"]["t"], list(ref.t), atol=5e-5)


def _icp_worker(rank, world, port, mode, q):
    for p in (os.path.join(ROOT, "pointclouds-rs_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist

    import oracle_ffi
    from cpu_engine import CpuEngine
    from pointclouds_rs import synth
    from pointclouds_rs.dist import Collectives, icp_sharded, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = oracle_ffi.load()
    eng = CpuEngine(o)
    coll = Collectives(dist)
    tgt = synth.hemisphere(1000, 7, 3.0)
    tn = o.normals(*tgt, 10, algo=1)
    src = o.apply_transform(*tgt, synth.rot_z(-0.03), (0.1, 0.05, -0.08))
    sb, se = shard_range(len(src[0]), world, rank)
    out = dict(rank=rank)
    for name, iters, tol, max_d in (("tol", 40, 1e-4, np.inf), ("bounded", 6, 0.0, 0.5), ("metrics", 0, 1e-5, np.inf)):
        r = icp_sharded(eng, coll, mode, [eng.from_numpy(c[sb:se]) for c in src], len(src[0]),
                        [eng.from_numpy(c) for c in tgt],
                        [eng.from_numpy(c) for c in tn] if mode == 1 else None, iters, tol, max_d)
        out[name] = dict(rot=list(r["rotation"]), t=list(r["translation"]), rmse=r["rmse"],
                         fitness=r["fitness"], iters=r["num_iterations"], converged=r["converged"])
    q.put(out)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", [0, 1], ids=["point_to_point", "point_to_plane"])
def test_sharded_icp_three_uneven_ranks(oracle, mode):
    """1000 source points over three ranks (334 / 333 / 333): a live tolerance, a bounded
    correspondence distance and the metrics-only pass of max_iterations = 0, every rank with the
    same answer and the loop control of the single-process oracle."""
    from pointclouds_rs import synth

    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_icp_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=120) for _ in procs], key=lambda d: d["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    tgt = synth.hemisphere(1000, 7, 3.0)
    tn = oracle.normals(*tgt, 10, algo=1)
    src = oracle.apply_transform(*tgt, synth.rot_z(-0.03), (0.1, 0.05, -0.08))
    twin = 1 if mode == 1 else 2  # f64 sums, as the sharded engine accumulates them
    for name, iters, tol, max_d in (("tol", 40, 1e-4, np.inf), ("bounded", 6, 0.0, 0.5), ("metrics", 0, 1e-5, np.inf)):
        rc, ref, _ = oracle.icp(twin, src, tgt, tn if mode == 1 else None, max_iter=iters, tol=tol,
                                max_d=max_d, algo=1)
        assert rc == 0
        assert outs[0][name] == outs[1][name] == outs[2][name], name
        got = outs[0][name]
        assert got["iters"] == ref.num_iterations and got["converged"] == bool(ref.converged), name
        assert got["fitness"] == ref.fitness, name
        assert abs(got["rmse"] - ref.rmse) <= 1e-4 * ref.rmse + 1e-6, name
        assert np.allclose(got["rot"], list(ref.rot), atol=2e-5), name
        assert np.allclose(got["t"], list(ref.t), atol=5e-5), name
