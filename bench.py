#!/usr/bin/env python3
"""Benchmark of the KNN hot path on MI355X.

Workload (BASELINE.json configs[1]): statistical_outlier_removal k=16 (std_mul 1.0) + PCA normals
k=16 on a synthetic uniform-random cloud, 1 M points per GPU, inputs resident in HBM.  A step is
one full pass: [all-gather of the column shards when N > 1] -> index build -> fused neighbour
search + SOR mean + normals -> [all-gather of the means] -> global threshold -> inlier mask.
N > 1 shards the queries of one N x 1 M-point cloud over the ranks (weak scaling).

Prints ONE JSON line on rank 0.  `roofline` prices the neighbour-search kernel with HIP events
recorded on its own stream; `cpu_baseline` times the CPU oracle (a C port of the reference
algorithms, reference threading: SOR single-threaded, normals on all cores) on the same cloud.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))

POINTS_PER_GPU = 1_000_000
K_SOR = 16
K_NRM = 16
STD_MUL = 1.0
BYTES_PER_POINT = 29  # 12 B xyz read + 4 B mean + 1 B mask + 12 B normal written
HBM_PEAK_GBPS = 8000.0


def cpu_baseline(x, y, z):
    """The oracle (test infrastructure) as the CPU leg: never part of the measured GPU path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi

    o = oracle_ffi.load()
    cores = o.max_threads()
    t0 = time.perf_counter()
    o.sor(x, y, z, K_SOR, STD_MUL, algo=1, threads=1)          # the reference's SOR loop is sequential
    t1 = time.perf_counter()
    o.normals(x, y, z, K_NRM, (0.0, 0.0, 0.0), algo=1, threads=cores)  # rayon par_iter
    t2 = time.perf_counter()
    n = len(x)
    return {
        "value": n / (t2 - t0) / 1e6,
        "unit": "Mpoints/s",
        "cores": cores,
        "kind": "port",
        "sample": f"the whole {n}-point cloud once: SOR k={K_SOR} on 1 thread ({t1 - t0:.2f} s, as the "
                  f"reference) + normals k={K_NRM} on {cores} threads ({t2 - t1:.2f} s); exact uniform-grid "
                  "search in place of the kiddo KD-tree",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=POINTS_PER_GPU, help="points per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true",
                    help="skip the side timings of the stages around the hot path")
    ap.add_argument("--icp-iters", type=int, default=50, help="ICP iterations for the side metric (0 = skip)")
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    # Rehearsal of the multi-rank code path on a box with ONE GPU (PCRS_BENCH_REHEARSE=1): every
    # rank uses cuda:0 and the collectives run over gloo through host memory.  The line it prints
    # exercises the control flow only and is marked as a rehearsal, never a measurement.
    rehearse = world > 1 and os.environ.get("PCRS_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: there is no CPU fallback of the product path")
    torch.cuda.set_device(local_rank)

    from pointclouds_rs import _ffi, synth
    from pointclouds_rs.dist import Collectives, HipEngine, shard_range, sor_normals_sharded

    engine = HipEngine(local_rank)
    coll = Collectives(dist, stage_through_host=rehearse)
    n_total = args.points * world
    begin, end = shard_range(n_total, world, rank)
    # one cloud of n_total points at unit density: side 100 * world^(1/3); each rank generates
    # only its own index range (the generator is a pure function of the index)
    side = 100.0 * (n_total / 1_000_000.0) ** (1.0 / 3.0)
    full_host = None
    cols = []
    for axis in range(3):
        u = synth.uniform01(42, n_total, axis)[begin:end] if world > 1 else synth.uniform01(42, n_total, axis)
        cols.append((u * np.float32(side)).astype(np.float32))
    if rank == 0 and world == 1:
        full_host = cols
    shard = [engine.from_numpy(c) for c in cols]
    nq = end - begin
    # With several ranks each one answers a run of whole cells of the replicated index (a compact
    # region of space) instead of the scattered points it was handed; the runs follow the cell
    # boundaries, so the buffers leave room for a little imbalance.
    ownership = "cells" if world > 1 else "index"
    cap = nq if world == 1 else nq + nq // 8 + 65536
    outputs = dict(mean=engine.empty(cap), keep=engine.empty(cap, "uint8"), nx=engine.empty(cap),
                   ny=engine.empty(cap), nz=engine.empty(cap))
    if world > 1:
        outputs.update(point=engine.empty(cap, "int32"), mean_by_index=engine.empty(n_total))

    def step(want_stats=False):
        return sor_normals_sharded(engine, coll, shard, n_total, K_SOR, STD_MUL, K_NRM,
                                   (0.0, 0.0, 0.0), outputs, want_stats=want_stats,
                                   ownership=ownership)

    search_stats = step(want_stats=True)["stats"]  # untimed: how many queries each stage answered
    for _ in range(args.warmup):
        res = step()
    torch.cuda.synchronize()
    coll.barrier()
    engine.timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    coll.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = engine.timing_read()
    engine.timing(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cpu" if rehearse else engine.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = n_total * args.steps / elapsed / 1e6
    avg_kernel_s = (kernel_ms / max(launches, 1)) * 1e-3
    nq = int(res["mean"].numel())  # queries this rank answered per step
    achieved = BYTES_PER_POINT * nq / avg_kernel_s / 1e9 if launches else 0.0

    icp_line = None
    if args.icp_iters > 0 and rank == 0:
        icp_line = bench_icp(engine, synth, args.icp_iters)
    stages_line = None
    host_line = None
    if not args.no_stages and rank == 0 and world == 1:
        stages_line = bench_stages(engine, shard, args.points)
        host_line = bench_host_call(full_host)

    if rank == 0:
        out = {
            "metric": "Mpoints/sec through SOR k=16 + normals on 1M-pt cloud; ICP iters/sec",
            "value": value,
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: all ranks on one GPU, gloo)",
            "config": {
                "workload": f"SOR k={K_SOR} std_mul={STD_MUL} + normals k={K_NRM}, uniform-random cloud, "
                            f"{args.points} points per GPU ({n_total} total), queries sharded over ranks",
                "points_per_gpu": args.points,
                "ownership": res["ownership"],
                "queries_rank0": nq,
                "n_keep_rank0": int(res["n_keep"]),
                "threshold": res["thr"],
                "search_stats_rank0": dict(zip(("cell_walk", "wave_retry", "brute"), search_stats)),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": None,
                "kernel": "knn_cell_kernel<17, true>: neighbour search over every query, fused with the SOR "
                          "mean and the covariance of the PCA normal",
                "kernel_ms_avg": avg_kernel_s * 1e3,
                "kernel_launches": launches,
                "algorithmic_bytes_per_launch": BYTES_PER_POINT * nq,
                "note": "the kernel is VALU-bound (distance + top-k selection), not HBM-bound: "
                        "29 B/pt of algorithmic traffic against thousands of lane-ops per point",
            },
        }
        if icp_line:
            out["icp"] = icp_line
        if stages_line:
            out["other_stages"] = stages_line
        if host_line:
            out["host_buffer_call"] = host_line
        out["roofline"]["traffic"] = pmc_traffic()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(*full_host)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic():
    """HBM bytes per launch of the neighbour-search kernel from the committed counter passes
    (profiles/, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this benchmark);
    None when the summary is not there.  FETCH_SIZE and WRITE_SIZE are in KiB; the kernel reads
    16-byte points per lane from scattered slots, an access pattern for which the counter is not
    calibrated, so the figure is reported as read."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_counters.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except OSError:
        return None
    for name, counters in d.items():
        if "knn_cell_kernel<17, true>" in name and "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
            return (counters["FETCH_SIZE"]["mean_per_launch"] + counters["WRITE_SIZE"]["mean_per_launch"]) * 1024.0
    return None


def bench_host_call(cols):
    """The same workload through the host-pointer entry point (what a Rust caller binds):
    pcrs_sor_normals with H2D of the columns and D2H of mask + normals inside the call.  Never
    the headline value; it shows what PCIe adds."""
    import numpy as np

    from pointclouds_rs import hip

    x, y, z = cols
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        keep, n_keep, thr, nrm = hip.sor_normals(x, y, z, K_SOR, STD_MUL, K_NRM)
        ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts[1:]))
    return {"entry_point": "pcrs_sor_normals (pageable host buffers)", "ms": dt * 1e3,
            "Mpoints_per_s": len(x) / dt / 1e6, "n_keep": int(n_keep)}


def bench_stages(engine, cols, n):
    """Side timings (not the headline metric) of the stages either side of the hot path on the
    same resident cloud: voxel_downsample, radius_outlier_removal, plane RANSAC behind its
    sampling, euclidean_cluster.  Wall time of the device-pointer entry points, median of 5."""
    import ctypes as C

    import numpy as np
    import torch

    lib, ctx = engine.lib, engine.ctx
    p = engine._p
    out = {}

    def timed(fn, reps=5):
        ts = []
        for _ in range(reps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts[1:]))

    ox, oy, oz = (engine.empty(n) for _ in range(3))
    nv = C.c_size_t(0)
    dt = timed(lambda: ctx.check(lib.pcrs_voxel_downsample_dev(ctx.handle, *[p(c) for c in cols], n, 1.0,
                                                              p(ox), p(oy), p(oz), C.byref(nv))))
    out["voxel_downsample"] = {"voxel_size": 1.0, "voxels": int(nv.value), "ms": dt * 1e3,
                               "Mpoints_per_s": n / dt / 1e6}
    keep = engine.empty(n, "uint8")
    nk = C.c_size_t(0)
    dt = timed(lambda: ctx.check(lib.pcrs_radius_outlier_dev(ctx.handle, *[p(c) for c in cols], n, 1.5, 12,
                                                            p(keep), None, C.byref(nk))))
    out["radius_outlier_removal"] = {"radius": 1.5, "min_neighbors": 12, "kept": int(nk.value),
                                     "ms": dt * 1e3, "Mpoints_per_s": n / dt / 1e6}
    from pointclouds_rs import sample_triples

    smp = engine.torch.from_numpy(sample_triples(n, 1000, 1).astype(np.int64).astype(np.int32)).to(engine.device)
    model = (C.c_float * 4)()
    nin = C.c_size_t(0)
    best = C.c_uint64(0)
    ns = smp.shape[0]
    dt = timed(lambda: ctx.check(lib.pcrs_ransac_plane_samples_dev(
        ctx.handle, *[p(c) for c in cols], n, p(smp), ns, 0.5, model, p(keep), C.byref(nin), C.byref(best))))
    out["ransac_plane"] = {"samples": int(ns), "threshold": 0.5, "inliers": int(nin.value), "ms": dt * 1e3,
                           "Gpoint_plane_tests_per_s": n * ns / dt / 1e9}
    offs = engine.empty(n + 1, "int32")
    idx = engine.empty(n, "int32")
    nc = C.c_size_t(0)
    ni = C.c_size_t(0)
    dt = timed(lambda: ctx.check(lib.pcrs_euclidean_cluster_dev(
        ctx.handle, *[p(c) for c in cols], n, 0.8, 2, n, p(offs), p(idx), C.byref(nc), C.byref(ni))))
    out["euclidean_cluster"] = {"threshold": 0.8, "min_size": 2, "clusters": int(nc.value),
                                "ms": dt * 1e3, "Mpoints_per_s": n / dt / 1e6}
    return out


def bench_icp(engine, synth, iters):
    """configs[2]: point-to-plane ICP, two 100 K-point scans, tolerance 0 so every iteration runs."""
    import numpy as np
    import torch

    from pointclouds_rs import _ffi

    n = 100_000
    tgt_h = synth.hemisphere(n, 99, 5.0)
    R = synth.rot_z(0.05)
    t = np.array([0.3, -0.2, 0.1], dtype=np.float32)
    P = np.stack(tgt_h, 1)
    S = (P @ R.T + t).astype(np.float32)
    tgt = [engine.from_numpy(c) for c in tgt_h]
    src = [engine.from_numpy(np.ascontiguousarray(S[:, a])) for a in range(3)]
    nrm = [engine.empty(n) for _ in range(3)]
    ix = engine.build_index(*tgt, 16)
    engine.sor_normals_range(ix, 0, n, 0, 16, (0.0, 0.0, 0.0), None, *nrm)
    engine.destroy_index(ix)
    h = engine.icp_create(src, tgt, nrm)
    engine.icp_run(h, _ffi.ICP_POINT_TO_PLANE, 3, 0.0, float("inf"))  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = engine.icp_run(h, _ffi.ICP_POINT_TO_PLANE, iters, 0.0, float("inf"))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    engine.icp_destroy(h)
    return {
        "workload": f"point-to-plane ICP, {n} x {n} points (hemisphere r=5, Rz(0.05)+t), {iters} iterations, tolerance 0",
        "iters_per_sec": res.num_iterations / dt,
        "ms_per_iter": dt / max(res.num_iterations, 1) * 1e3,
        "rmse": res.rmse,
        "fitness": res.fitness,
        "num_iterations": int(res.num_iterations),
    }


if __name__ == "__main__":
    main()
