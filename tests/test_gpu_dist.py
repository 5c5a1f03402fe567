"""Two ranks driving the real HIP engine (both on cuda:0, collectives over gloo staged through
host memory): pointclouds_rs.dist with both kinds of ownership against the oracle.  The RCCL
path itself needs one GPU per rank and runs in bench.py --gpus N."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
F = np.float32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cloud():
    from pointclouds_rs import synth

    return synth.kitti_scene(3, 51001, 2000, 1000)  # 58001 points: uneven shards


def _worker(rank, world, port, q, icp_inputs):
    for p in (os.path.join(ROOT, "pointclouds-rs_amd"), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    from pointclouds_rs.dist import Collectives, HipEngine, shard_range, sor_normals_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = HipEngine.shared(0)
    coll = Collectives(dist, stage_through_host=True)  # gloo has no device collectives here
    x, y, z = _cloud()
    n_total = len(x)
    b, e = shard_range(n_total, world, rank)
    shard = [eng.from_numpy(c[b:e]) for c in (x, y, z)]
    out = dict(rank=rank, begin=b, end=e)
    for own in ("index", "cells", "slabs"):
        r = sor_normals_sharded(eng, coll, shard, n_total, 16, 1.0, 16, ownership=own)
        torch.cuda.synchronize()
        out[own] = dict(ownership=r["ownership"], thr=r["thr"], n_keep=r["n_keep"],
                        escalated=r.get("escalated", 0),
                        point=None if r["point"] is None else r["point"].cpu().numpy().view(np.uint32).copy(),
                        keep=r["keep"].cpu().numpy().copy(), mean=r["mean"].cpu().numpy().copy(),
                        normals=[r[c].cpu().numpy().copy() for c in ("nx", "ny", "nz")])
    # sharded ICP (point-to-plane): source split over the ranks, target and normals replicated
    from pointclouds_rs import _ffi, synth
    from pointclouds_rs.dist import icp_sharded

    # the parent made the inputs with the oracle and checks against the oracle's run on them
    inp = np.load(icp_inputs)
    tgt = [inp["t%d" % a] for a in range(3)]
    tn = [inp["n%d" % a] for a in range(3)]
    S = np.stack([inp["s%d" % a] for a in range(3)], 1)
    sb, se = shard_range(len(S), world, rank)
    r = icp_sharded(eng, coll, _ffi.ICP_POINT_TO_PLANE,
                    [eng.from_numpy(np.ascontiguousarray(S[sb:se, a])) for a in range(3)], len(S),
                    [eng.from_numpy(c) for c in tgt], [eng.from_numpy(c) for c in tn], 12, 0.0)
    out["icp"] = dict(rot=list(r["rotation"]), t=list(r["translation"]), rmse=r["rmse"],
                      fitness=r["fitness"], iters=r["num_iterations"])
    # the same with a live tolerance: the loop must end in the reference's iteration
    r = icp_sharded(eng, coll, _ffi.ICP_POINT_TO_PLANE,
                    [eng.from_numpy(np.ascontiguousarray(S[sb:se, a])) for a in range(3)], len(S),
                    [eng.from_numpy(c) for c in tgt], [eng.from_numpy(c) for c in tn], 30, 1e-4)
    out["icp_tol"] = dict(rot=list(r["rotation"]), t=list(r["translation"]), rmse=r["rmse"],
                          fitness=r["fitness"], iters=r["num_iterations"],
                          converged=r["converged"])
    q.put(out)
    dist.destroy_process_group()


def _icp_inputs(oracle):
    from pointclouds_rs import synth

    tgt = synth.hemisphere(4000, 99, 5.0)
    tn = oracle.normals(*tgt, 12, algo=1)
    src = oracle.apply_transform(*tgt, synth.rot_z(0.04), (0.2, -0.1, 0.05))
    return src, tgt, tn


@pytest.mark.timeout(300)
def test_two_ranks_both_ownerships(oracle, tmp_path):
    import queue

    world = 2
    src, tgt, tn = _icp_inputs(oracle)
    icp_inputs = str(tmp_path / "icp_inputs.npz")
    np.savez(icp_inputs, **{"s%d" % a: src[a] for a in range(3)},
             **{"t%d" % a: tgt[a] for a in range(3)}, **{"n%d" % a: tn[a] for a in range(3)})
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, icp_inputs)) for r in range(world)]
    for p in procs:
        p.start()
    outs = []
    try:
        for _ in range(200):  # a rank that dies must not leave the test waiting
            try:
                outs.append(q.get(timeout=1.0))
            except queue.Empty:
                assert all(p.exitcode in (None, 0) for p in procs), [p.exitcode for p in procs]
            if len(outs) == world:
                break
        assert len(outs) == world
    finally:
        for p in procs:
            p.join(20)
            if p.is_alive():
                p.kill()
    outs.sort(key=lambda d: d["rank"])
    x, y, z = _cloud()
    n_total = len(x)
    keep, n_keep, thr, mean = oracle.sor(x, y, z, 16, 1.0, algo=1)
    rn = oracle.normals(x, y, z, 16, algo=1)
    for own in ("index", "cells", "slabs"):
        assert sum(o[own]["n_keep"] for o in outs) == n_keep
        pts = []
        for o in outs:
            r = o[own]
            assert r["ownership"] == own
            pt = np.arange(o["begin"], o["end"]) if own == "index" else r["point"]
            pts.append(pt)
            assert F(r["thr"]).view(np.uint32) == F(thr).view(np.uint32)
            assert np.array_equal(r["mean"].view(np.uint32), mean[pt].view(np.uint32))
            assert np.array_equal(r["keep"].astype(bool), keep[pt])
            for got, ref in zip(r["normals"], rn):
                assert np.abs(got.astype(np.float64) - ref[pt]).max() <= 1e-5
        assert np.array_equal(np.sort(np.concatenate(pts)), np.arange(n_total))
    # ICP: both ranks hold the same transform, equal to the oracle's single-process run
    rc, ref, _ = oracle.icp(1, src, tgt, tn, max_iter=12, tol=0.0, algo=1)
    assert outs[0]["icp"]["rot"] == outs[1]["icp"]["rot"] and outs[0]["icp"]["t"] == outs[1]["icp"]["t"]
    for o in outs:
        assert o["icp"]["iters"] == ref.num_iterations == 12
        assert o["icp"]["fitness"] == ref.fitness
        assert o["icp"]["rmse"] == ref.rmse
        assert np.allclose(o["icp"]["rot"], list(ref.rot), rtol=0, atol=2.5e-7)
        assert np.allclose(o["icp"]["t"], list(ref.t), rtol=0, atol=2.5e-7)
    rc, ref, _ = oracle.icp(1, src, tgt, tn, max_iter=30, tol=1e-4, algo=1)
    assert 1 < ref.num_iterations < 30 and ref.converged
    assert outs[0]["icp_tol"] == outs[1]["icp_tol"]  # bit for bit the same on both ranks
    for o in outs:
        assert o["icp_tol"]["iters"] == ref.num_iterations
        assert o["icp_tol"]["converged"] == bool(ref.converged)
        assert o["icp_tol"]["fitness"] == ref.fitness
        assert o["icp_tol"]["rmse"] == ref.rmse  # the reference's own f32 sum, gathered over the ranks
        assert np.allclose(o["icp_tol"]["rot"], list(ref.rot), rtol=0, atol=2.5e-7)
        assert np.allclose(o["icp_tol"]["t"], list(ref.t), rtol=0, atol=2.5e-7)
