#!/usr/bin/env python3
"""Benchmark of the KNN hot path on MI355X.

Default workload (BASELINE.json configs[1], `--config c2`): statistical_outlier_removal k=16
(std_mul 1.0) + PCA normals k=16 on a synthetic uniform-random cloud, 1 M points per GPU, inputs
resident in HBM.  A step is one full pass: [all-gather of the column shards when N > 1] -> index
build -> fused neighbour search + SOR mean + normals -> [exchange of the means] -> global
threshold -> inlier mask.  N > 1 shards the queries of one N x 1 M-point cloud over the ranks
(weak scaling).

`python bench.py --gpus N` with N > 1 starts its own ranks (one process per GPU through
torch.distributed.run, before anything touches a GPU); started under torch.distributed.run it
takes RANK / LOCAL_RANK / WORLD_SIZE from the environment.  Rank 0 prints ONE JSON line.

Other workloads (side measurements, never the default line):
  --config c3   point-to-plane ICP of two 100 K-point scans, 50 iterations; N > 1 shards the source
  --config c4   one 100 M-point aerial-LiDAR shaped tile, SOR + normals, queries sharded over N GPUs
  --config c5   N KITTI-like frames through voxel -> SOR -> RANSAC -> clusters, one frame per GPU

`roofline` prices the neighbour-search kernel with HIP events recorded on its own stream;
`cpu_baseline` times the CPU oracle (a C port of the reference algorithms) on the same cloud.
"""
import argparse
import json
import os
import socket
import subprocess
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
# f32 vector peak of the chip: 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-operations
# per second for ordinary instructions; packed f32 (v_pk_*: two per lane) doubles it to 78.6 T,
# which as FMA is the 157 TFLOP/s of the data sheet
VALU_PEAK_TOPS = 78.6
N_XCD, N_SIMD = 8, 1024  # MI355X: 8 XCDs, 256 CUs x 4 SIMDs
METRIC = "Mpoints/sec through SOR k=16 + normals on 1M-pt cloud; ICP iters/sec"
PROFILE_TAG = "r03"
SEARCH_KERNEL = "knn_tag_kernel<17>"  # the dominant kernel of the default workload


# --------------------------------------------------------------------------------- CPU legs

def _oracle():
    """The oracle (test infrastructure) as the CPU leg: never part of the measured GPU path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi

    return oracle_ffi.load()


def cpu_baseline(x, y, z, synth):
    """Reference threading ("faithful": SOR on one thread, normals on all) is the reported value;
    the all-thread variant, the reference's own benchmark size (100 K points, k = 10) and an ICP
    leg are listed beside it.  Everything is a bounded sample of about 10 s of CPU work."""
    import numpy as np

    o = _oracle()
    cores = o.max_threads()
    n = len(x)
    t0 = time.perf_counter()
    o.sor(x, y, z, K_SOR, STD_MUL, algo=1, threads=1)          # the reference's SOR loop is sequential
    t1 = time.perf_counter()
    o.normals(x, y, z, K_NRM, (0.0, 0.0, 0.0), algo=1, threads=cores)  # rayon par_iter
    t2 = time.perf_counter()
    o.sor(x, y, z, K_SOR, STD_MUL, algo=1, threads=cores)
    t3 = time.perf_counter()
    # C1: the reference's own benchmark (benches/bench_filters.rs:30-41), SOR k=10 on 100 K points
    cx, cy, cz = synth.uniform_cloud(100_000, 42)
    t4 = time.perf_counter()
    o.sor(cx, cy, cz, 10, 1.0, algo=1, threads=1)
    t5 = time.perf_counter()
    # ICP: point-to-plane, 20 K x 20 K points, 10 iterations on one thread (the reference's loop
    # is sequential); per-iteration time scales with the source size
    m = 20_000
    tgt = synth.hemisphere(m, 99, 5.0)
    tn = o.normals(*tgt, 16, algo=1, threads=cores)
    src = o.apply_transform(*tgt, synth.rot_z(0.05), (0.3, -0.2, 0.1))
    t6 = time.perf_counter()
    rc, res, _ = o.icp(1, src, tgt, tn, max_iter=10, tol=0.0, algo=1, threads=1)
    t7 = time.perf_counter()
    return {
        "value": n / (t2 - t0) / 1e6,
        "unit": "Mpoints/s",
        "cores": cores,
        "kind": "port",
        "sample": f"the whole {n}-point cloud once: SOR k={K_SOR} on 1 thread ({t1 - t0:.2f} s, as the "
                  f"reference) + normals k={K_NRM} on {cores} threads ({t2 - t1:.2f} s); exact uniform-grid "
                  "search in place of the kiddo KD-tree",
        "variants": {
            "all_threads": {"value": n / ((t3 - t2) + (t2 - t1)) / 1e6, "unit": "Mpoints/s", "cores": cores,
                            "sample": f"SOR on {cores} threads ({t3 - t2:.2f} s) + normals on {cores} threads"},
            "c1_sor_k10_100k": {"value": 100_000 / (t5 - t4) / 1e6, "unit": "Mpoints/s", "cores": 1,
                                "ms": (t5 - t4) * 1e3,
                                "sample": "SOR k=10 std_mul=1.0 on 100 K uniform points, one thread "
                                          "(BENCHMARKS.md of the reference: 128 ms on an M4 Max)"},
            "icp_point_to_plane": {"value": int(res.num_iterations) / (t7 - t6), "unit": "iterations/s",
                                   "cores": 1, "points": m,
                                   "sample": f"{int(res.num_iterations)} iterations on {m} x {m} points, one thread; "
                                             f"scaled to 100 K points: {int(res.num_iterations) / (t7 - t6) * m / 100_000:.1f} it/s"},
        },
    }


# ------------------------------------------------------------------------------- launching

def launch_ranks(args):
    """`python bench.py --gpus N` as the driver calls it: this process touches no GPU and starts
    one rank per GPU through torch.distributed.run; the ranks' output passes through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def setup_ranks(args):
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    return world, rank, local_rank, rehearse, dist


def max_over_ranks(dist, torch, engine, rehearse, seconds):
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if rehearse else engine.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ------------------------------------------------------------------- c2 / c4: SOR + normals

def run_sor_normals(args):
    import numpy as np
    import torch

    world, rank, local_rank, rehearse, dist = setup_ranks(args)
    from pointclouds_rs import synth
    from pointclouds_rs.dist import Collectives, HipEngine, shard_range, sor_normals_sharded

    engine = HipEngine(local_rank)
    coll = Collectives(dist, stage_through_host=rehearse)
    c4 = args.config == "c4"
    if c4:
        # one tile of fixed size, its queries divided among the ranks (strong scaling)
        n_total = args.points if args.points else 100_000_000
        scale = (500.0, 500.0, 20.0)  # an aerial-LiDAR shaped slab at about 20 points per unit^3
        f = (n_total / 100_000_000.0) ** (1.0 / 3.0)
        scale = tuple(s * f for s in scale)
        seed = 9
        vp = (scale[0] / 2, scale[1] / 2, 1000.0)  # the sensor above the tile
    else:
        per = args.points if args.points else POINTS_PER_GPU
        n_total = per * world
        # one cloud of n_total points at unit density: side 100 * world^(1/3)
        side = 100.0 * (n_total / 1_000_000.0) ** (1.0 / 3.0)
        scale = (side, side, side)
        seed = 42
        vp = (0.0, 0.0, 0.0)
    begin, end = shard_range(n_total, world, rank)
    # each rank generates only its own index range (the generator is a pure function of the index;
    # the whole stream is materialised for one axis at a time)
    cols = []
    for axis in range(3):
        u = synth.uniform01(seed, n_total, axis)
        cols.append((u[begin:end] * np.float32(scale[axis])).astype(np.float32))
        del u
    full_host = cols if (rank == 0 and world == 1) else None
    shard = [engine.from_numpy(c) for c in cols]
    nq = end - begin
    # With several ranks each one indexes and answers a slab of space (equal point counts, cut along
    # the cloud's longest axis) instead of the scattered points it was handed; the cuts fall on the
    # edges of histogram bins, so the buffers leave room for a little imbalance.
    ownership = "slabs" if world > 1 else "index"
    cap = nq if world == 1 else nq + nq // 8 + 65536
    outputs = dict(mean=engine.empty(cap), keep=engine.empty(cap, "uint8"), nx=engine.empty(cap),
                   ny=engine.empty(cap), nz=engine.empty(cap))
    if world > 1:
        outputs.update(point=engine.empty(cap, "int32"), mean_by_index=engine.empty(n_total))

    # --dist-driver native: the library's own driver (pcrs_dist_sor_normals_dev, collectives issued
    # on RCCL inside the call) over a communicator of its own; the id travels over torch's.
    native = args.dist_driver == "native" and world > 1 and not rehearse
    comm = None
    if native:
        blob = [engine.dist_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(blob, src=0)
        comm = engine.dist_comm_init(blob[0], rank, world)
        native_out = {k: outputs[k] for k in ("point", "mean", "keep", "nx", "ny", "nz")}

    def step(want_stats=False):
        if native and not want_stats:
            return engine.dist_sor_normals(comm, rank, world, shard, n_total, K_SOR, STD_MUL, K_NRM,
                                           vp, outputs=native_out)
        return sor_normals_sharded(engine, coll, shard, n_total, K_SOR, STD_MUL, K_NRM, vp, outputs,
                                   want_stats=want_stats, ownership=ownership)

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
    elapsed = max_over_ranks(dist, torch, engine, rehearse, elapsed)

    ms_per_step = elapsed / args.steps * 1e3
    value = n_total * args.steps / elapsed / 1e6
    avg_kernel_s = (kernel_ms / max(launches, 1)) * 1e-3
    nq = int(res["mean"].numel())  # queries this rank answered per step
    achieved = BYTES_PER_POINT * nq / avg_kernel_s / 1e9 if launches else 0.0

    # Side lines on one GPU: the SOR stage alone (the stage north_star's target is quoted on) and
    # the same step with nothing left in the caches from the step before.
    sor_only_line = cold_line = None
    if world == 1 and not c4 and not args.no_stages:
        sor_only_line = bench_sor_only(engine, coll, shard, n_total, vp, outputs, args)
        cold_line = bench_cold(engine, step, args)

    aerial_line = None
    if c4 and world == 1 and not args.no_stages:
        aerial_line = bench_aerial(engine, coll, n_total, args)

    icp_line = stages_line = host_line = None
    if not c4 and rank == 0:
        if args.icp_iters > 0:
            icp_line = bench_icp(engine, synth, args.icp_iters)
        if not args.no_stages and world == 1:
            stages_line = bench_stages(engine, shard, nq)
            host_line = bench_host_call(full_host)

    if rank == 0:
        shape = ("aerial-LiDAR shaped tile" if c4 else "uniform-random cloud")
        out = {
            "metric": METRIC if not c4 else "Mpoints/sec through SOR k=16 + normals on one 100M-pt tile",
            "value": value,
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if c4 else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: all ranks on one GPU, gloo)",
            "config": {
                "workload": f"SOR k={K_SOR} std_mul={STD_MUL} + normals k={K_NRM}, {shape}, "
                            f"{n_total} points in all ({n_total // world} per GPU), queries sharded over ranks",
                "name": args.config,
                "points_per_gpu": n_total // world,
                "ownership": res["ownership"],
                "driver": "libpcrs_hip (pcrs_dist_sor_normals_dev)" if native else "pointclouds_rs.dist",
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
                "kernel": SEARCH_KERNEL + ": neighbour search over every query (one walk of the 27-cell "
                          "cube, insertion chain over distance keys tagged with the candidate's ordinal), "
                          "fused with the SOR mean and the covariance of the PCA normal",
                "kernel_ms_avg": avg_kernel_s * 1e3,
                "kernel_launches": launches,
                "algorithmic_bytes_per_launch": BYTES_PER_POINT * nq,
                "note": "nominally HBM-bound (29 B/pt), in fact bound by vector-ALU issue: the SIMDs are "
                        "occupied for `valu.issue_busy` of the launch by unpacked f32 instructions",
            },
        }
        # the committed counter passes were taken on the default workload (1 M points per GPU)
        if not c4 and n_total // world == POINTS_PER_GPU:
            out["roofline"].update(profile_counters(avg_kernel_s))
        if aerial_line:
            out["aerial_scene"] = aerial_line
        if sor_only_line:
            out["sor_only"] = sor_only_line
        if cold_line:
            out["cold"] = cold_line
        if icp_line:
            out["icp"] = icp_line
        if stages_line:
            out["other_stages"] = stages_line
        if host_line:
            out["host_buffer_call"] = host_line
        if world == 1 and not c4 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(*full_host, synth)
        print(json.dumps(out), flush=True)
    if comm is not None:
        engine.dist_comm_destroy(comm)
    if world > 1:
        dist.destroy_process_group()
    return 0


def bench_aerial(engine, coll, n_target, args):
    """The same step on a scene shaped like the reference's aerial example (terrain surface,
    buildings, trees: examples/python/aerial_lidar.py:26-137) instead of the uniform slab, scaled to
    about n_target points; one GPU holds the whole scene."""
    import torch

    from pointclouds_rs import synth
    from pointclouds_rs.dist import sor_normals_sharded

    scale = n_target / 2_410_000.0
    cols = synth.aerial_scene(scale, 42)
    n = len(cols[0])
    shard = [engine.from_numpy(c) for c in cols]
    del cols
    vp = (250.0, 250.0, 1000.0)
    outputs = dict(mean=engine.empty(n), keep=engine.empty(n, "uint8"), nx=engine.empty(n),
                   ny=engine.empty(n), nz=engine.empty(n))

    def step(want_stats=False):
        return sor_normals_sharded(engine, coll, shard, n, K_SOR, STD_MUL, K_NRM, vp, outputs,
                                   want_stats=want_stats)

    stats = step(want_stats=True)["stats"]
    steps = max(3, min(args.steps, 10))
    for _ in range(2):
        res = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"workload": f"SOR k={K_SOR} + normals k={K_NRM} on an aerial scene of {n} points (terrain, five "
                        "buildings, eight trees on 500 m x 500 m)", "ms_per_step": dt * 1e3,
            "Mpoints_per_s": n / dt / 1e6, "steps": steps, "n_keep": int(res["n_keep"]),
            "search_stats": dict(zip(("cell_walk", "wave_retry", "brute"), stats))}


def bench_sor_only(engine, coll, shard, n_total, vp, outputs, args):
    """statistical_outlier_removal alone (no normals) on the same resident cloud: index build,
    search for distances only (knn_cell_kernel<17,false>: one walk, no neighbour identities),
    statistics, mask.  17 algorithmic bytes per point (12 read, 4 mean, 1 mask)."""
    import torch

    from pointclouds_rs.dist import sor_normals_sharded

    def step():
        return sor_normals_sharded(engine, coll, shard, n_total, K_SOR, STD_MUL, 0, vp, outputs)

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    engine.timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kernel_ms, launches = engine.timing_read()
    engine.timing(False)
    ksec = kernel_ms / max(launches, 1) * 1e-3
    gbps = 17 * n_total / ksec / 1e9 if launches else 0.0
    return {"workload": f"SOR k={K_SOR} std_mul={STD_MUL} alone, {n_total} points",
            "ms_per_step": dt / args.steps * 1e3, "Mpoints_per_s": n_total * args.steps / dt / 1e6,
            "n_keep": int(res["n_keep"]), "kernel": "knn_cell_kernel<17,false>",
            "kernel_ms_avg": ksec * 1e3, "algorithmic_bytes_per_launch": 17 * n_total,
            "achieved_GBps": gbps, "hbm_frac": gbps / HBM_PEAK_GBPS}


def bench_cold(engine, step, args, scrub_bytes=1 << 30):
    """The default step with cold caches: between two steps a 1 GiB buffer is rewritten (four
    times the 256 MB of memory-side cache, far beyond the L2s), so the cloud, the index and every
    intermediate of the step before are gone and the step's reads come from HBM.  Each step is
    timed on its own with events on the stream the library runs on; the scrub is not timed."""
    import torch

    scrub = torch.empty(scrub_bytes // 4, dtype=torch.float32, device=engine.device)
    steps = min(args.steps, 10)
    engine.timing(True)
    pairs = []
    for i in range(steps + 1):
        scrub.fill_(float(i))
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    kernel_ms, launches = engine.timing_read()
    engine.timing(False)
    ms = sorted(a.elapsed_time(b) for a, b in pairs[1:])
    del scrub
    return {"method": "1 GiB rewritten between steps (untimed); each step timed by events on its stream",
            "steps": steps, "ms_per_step_median": ms[len(ms) // 2], "ms_per_step_min": ms[0],
            "ms_per_step_max": ms[-1], "kernel_ms_avg": kernel_ms / max(launches, 1),
            "Mpoints_per_s": POINTS_PER_GPU / (ms[len(ms) // 2] * 1e-3) / 1e6}


def profile_counters(avg_kernel_s):
    """HBM bytes and vector-ALU work per launch of the neighbour-search kernel from the committed
    counter passes of this benchmark (profiles/<tag>_pmc_counters.json: rocprofv3 --pmc, one
    pass per counter group; scripts/profile_bench.sh).  The file records the commit it was taken
    at: figures from another state of the kernels are marked stale.  FETCH_SIZE and WRITE_SIZE
    are in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes for wide coalesced
    reads, so the read side is doubled (an upper bound for this kernel's per-lane 16-byte loads)."""
    out = {"traffic": None}
    for tag in (PROFILE_TAG,):
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_counters.json")
        try:
            with open(path) as f:
                d = json.load(f)
        except OSError:
            continue
        meta = d.get("_meta", {})
        for name, c in d.items():
            if SEARCH_KERNEL not in name:
                continue
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                rd = c["FETCH_SIZE"]["mean_per_launch"] * 1024.0
                wr = c["WRITE_SIZE"]["mean_per_launch"] * 1024.0
                out["traffic"] = 2.0 * rd + wr
                out["traffic_detail"] = {"FETCH_SIZE_bytes": rd, "WRITE_SIZE_bytes": wr,
                                         "correction": "2 x FETCH_SIZE + WRITE_SIZE (gfx950)",
                                         "profile": os.path.basename(path),
                                         "profile_commit": meta.get("commit"),
                                         "profile_kernel_ms": meta.get("kernel_ms")}
            if "SQ_INSTS_VALU" in c and "SQ_THREAD_CYCLES_VALU" in c and avg_kernel_s > 0:
                insts = c["SQ_INSTS_VALU"]["mean_per_launch"]
                lane_ops = insts * 64.0
                # the counters come from the profiled build; the rate uses that build's kernel time
                # when the file has it, else this run's
                ksec = (meta.get("kernel_ms") or avg_kernel_s * 1e3) * 1e-3
                v = {"lane_ops": lane_ops, "achieved_Tops": lane_ops / ksec / 1e12,
                     "peak": VALU_PEAK_TOPS, "frac": lane_ops / ksec / 1e12 / VALU_PEAK_TOPS,
                     "lane_util": c["SQ_THREAD_CYCLES_VALU"]["mean_per_launch"] / insts / 64.0,
                     "profile": os.path.basename(path), "profile_commit": meta.get("commit")}
                if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
                    v["wave_wait_frac"] = c["SQ_WAIT_ANY"]["mean_per_launch"] / c["SQ_WAVE_CYCLES"]["mean_per_launch"]
                if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
                    # How long the vector ALUs were occupied.  SQ_ACTIVE_INST_VALU counts
                    # quad-cycles (an unpacked f32 instruction holds its SIMD for one), summed over
                    # the 1024 SIMDs; GRBM_GUI_ACTIVE counts the launch's cycles once per XCD.  `peak`
                    # above is the packed-f32 rate (two lane-ops per lane and quad-cycle): a kernel
                    # of unpacked instructions, like this one's v_med3 chain, tops out at half.
                    quads = c["GRBM_GUI_ACTIVE"]["mean_per_launch"] / N_XCD / 4.0
                    v["issue_busy"] = c["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / (N_SIMD * quads)
                    v["peak_unpacked"] = VALU_PEAK_TOPS / 2.0
                    v["frac_unpacked"] = v["achieved_Tops"] / (VALU_PEAK_TOPS / 2.0)
                out["valu"] = v
            return out
    return out


def bench_host_call(cols):
    """The same workload through the host-pointer entry point (what a Rust caller binds):
    pcrs_sor_normals with H2D of the columns and D2H of mask + normals inside the call.  Never
    the headline value; it shows what PCIe adds."""
    import numpy as np

    from pointclouds_rs import hip

    x, y, z = cols
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        keep, n_keep, thr, nrm = hip.sor_normals(x, y, z, K_SOR, STD_MUL, K_NRM)
        ts.append(time.perf_counter() - t0)
    dt = float(np.median(ts[1:]))
    moved = len(x) * 25  # 12 B/pt in, 1 + 12 B/pt out
    return {"entry_point": "pcrs_sor_normals (pageable host buffers)", "ms": dt * 1e3,
            "Mpoints_per_s": len(x) / dt / 1e6, "n_keep": int(n_keep),
            "pcie_bytes": moved, "pcie_GBps_effective": moved / dt / 1e9}


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


# ------------------------------------------------------------------------------- c3: ICP

def icp_scans(engine, synth, n):
    """configs[2]: target = n points on a hemisphere of radius 5 with its own normals (k = 16),
    source = the target moved by Rz(0.05) and (0.3, -0.2, 0.1)."""
    import numpy as np

    tgt_h = synth.hemisphere(n, 99, 5.0)
    R = synth.rot_z(0.05)
    t = np.array([0.3, -0.2, 0.1], dtype=np.float32)
    P = np.stack(tgt_h, 1)
    S = (P @ R.T + t).astype(np.float32)
    tgt = [engine.from_numpy(c) for c in tgt_h]
    src_h = [np.ascontiguousarray(S[:, a]) for a in range(3)]
    nrm = [engine.empty(n) for _ in range(3)]
    ix = engine.build_index(*tgt, 16)
    engine.sor_normals_range(ix, 0, n, 0, 16, (0.0, 0.0, 0.0), None, *nrm)
    engine.destroy_index(ix)
    return src_h, tgt, nrm


def bench_icp(engine, synth, iters, n=100_000):
    """Point-to-plane ICP on one GPU: tolerance 0 so that every iteration runs, then the
    reference's default tolerance 1e-5 (the loop ends when the rmse settles)."""
    import torch

    from pointclouds_rs import _ffi

    src_h, tgt, nrm = icp_scans(engine, synth, n)
    src = [engine.from_numpy(c) for c in src_h]
    h = engine.icp_create(src, tgt, nrm)
    engine.icp_run(h, _ffi.ICP_POINT_TO_PLANE, 3, 0.0, float("inf"))  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = engine.icp_run(h, _ffi.ICP_POINT_TO_PLANE, iters, 0.0, float("inf"))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    engine.icp_run(h, _ffi.ICP_POINT_TO_PLANE, iters, 1e-5, float("inf"))  # warm-up of the exact-rmse path
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    rtol = engine.icp_run(h, _ffi.ICP_POINT_TO_PLANE, iters, 1e-5, float("inf"))
    torch.cuda.synchronize()
    dtol = time.perf_counter() - t1
    engine.icp_destroy(h)
    return {
        "workload": f"point-to-plane ICP, {n} x {n} points (hemisphere r=5, Rz(0.05)+t), {iters} iterations, tolerance 0",
        "iters_per_sec": res.num_iterations / dt,
        "ms_per_iter": dt / max(res.num_iterations, 1) * 1e3,
        "rmse": res.rmse,
        "fitness": res.fitness,
        "num_iterations": int(res.num_iterations),
        "default_tolerance": {"tolerance": 1e-5, "num_iterations": int(rtol.num_iterations),
                              "converged": bool(rtol.converged), "ms_total": dtol * 1e3,
                              "ms_per_iter": dtol / max(rtol.num_iterations, 1) * 1e3, "rmse": rtol.rmse},
    }


def run_icp(args):
    import torch

    world, rank, local_rank, rehearse, dist = setup_ranks(args)
    from pointclouds_rs import _ffi, synth
    from pointclouds_rs.dist import Collectives, HipEngine, icp_sharded, shard_range

    engine = HipEngine(local_rank)
    coll = Collectives(dist, stage_through_host=rehearse)
    n = args.points if args.points else 100_000
    iters = args.icp_iters if args.icp_iters > 0 else 50
    if world == 1:
        line = bench_icp(engine, synth, iters, n)
        value, ms = line["iters_per_sec"], line["ms_per_iter"]
        extra = line
    else:
        src_h, tgt, nrm = icp_scans(engine, synth, n)
        b, e = shard_range(n, world, rank)
        shard = [engine.from_numpy(c[b:e]) for c in src_h]
        icp_sharded(engine, coll, _ffi.ICP_POINT_TO_PLANE, shard, n, tgt, nrm, 3, 0.0)  # warm-up
        torch.cuda.synchronize()
        coll.barrier()
        t0 = time.perf_counter()
        r = icp_sharded(engine, coll, _ffi.ICP_POINT_TO_PLANE, shard, n, tgt, nrm, iters, 0.0)
        torch.cuda.synchronize()
        coll.barrier()
        dt = max_over_ranks(dist, torch, engine, rehearse, time.perf_counter() - t0)
        value, ms = r["num_iterations"] / dt, dt / max(r["num_iterations"], 1) * 1e3
        extra = {"rmse": r["rmse"], "fitness": r["fitness"], "num_iterations": r["num_iterations"]}
    if rank == 0:
        print(json.dumps({
            "metric": "ICP iters/sec, point-to-plane, two 100K-point scans", "value": value,
            "unit": "iterations/s", "n_gpus": world, "steps": iters, "warmup": 3, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32/f64",
            "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: all ranks on one GPU, gloo)",
            "config": {"workload": f"point-to-plane ICP, {n} x {n} points, {iters} iterations, tolerance 0, "
                                   "source sharded over ranks, target replicated", "name": "c3"},
            "detail": extra}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------------- c5: frame pipeline

def run_frames(args):
    """configs[4]: the obstacle pipeline of examples/python/kitti_obstacle_detection.py:87-124
    (voxel 0.15 -> SOR k=20 x2.0 -> RANSAC ground 0.15 / 500 -> drop ground -> clusters 0.8 / 10 /
    20000), one synthetic frame per rank, no communication between ranks (replicas only)."""
    import numpy as np
    import torch

    world, rank, local_rank, rehearse, dist = setup_ranks(args)
    import pointclouds_rs as pc
    from pointclouds_rs import hip, synth
    from pointclouds_rs.dist import Collectives, HipEngine

    engine = HipEngine(local_rank, use_torch_stream=False)
    coll = Collectives(dist, stage_through_host=rehearse)
    x, y, z = synth.kitti_scene(42 + rank)
    dev = local_rank

    def frame_one_call():
        out = hip.obstacle_pipeline(x, y, z, 0.15, 20, 2.0, 0.15, 500, 1234, 0.8, 10, 20000,
                                    want_obstacles=False, device=dev)
        return out["n_voxels"], out["n_kept"], out["n_ground"], len(out["clusters"])

    def frame_by_stages():
        g = hip.voxel_downsample(x, y, z, 0.15, device=dev)
        keep, n_keep, thr, _ = hip.sor(*g, 20, 2.0, device=dev)
        c = [col[keep] for col in g]
        smp = pc.sample_triples(len(c[0]), 500, 1234)
        nrm, d, inl, best = hip.ransac_plane_samples(*c, smp, 0.15, device=dev)
        mask = np.ones(len(c[0]), dtype=bool)
        mask[inl] = False
        obst = [col[mask] for col in c]
        cl = hip.euclidean_cluster(*obst, 0.8, 10, 20000, device=dev)
        return len(g[0]), int(n_keep), len(inl), len(cl)

    # the stages one by one through host buffers (as a caller of the reference's API runs them),
    # as a side figure; the frame itself goes through pcrs_obstacle_pipeline
    info_stages, dt_stages = None, None
    if not args.no_stages:
        for _ in range(max(args.warmup, 1)):
            info_stages = frame_by_stages()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            info_stages = frame_by_stages()
        dt_stages = time.perf_counter() - t0
    frame = frame_one_call
    # The HIP runtime grows some of its pools once, a few ten thousand launches into a process: two
    # frames of 6 and 40 ms around the 200th (scripts/exp_c5_spikes.py), none in the 1 300 after.
    # A run of 20 timed frames either holds them or does not; 300 untimed frames first put every
    # run behind them, as any sensor stream is after a third of a second.
    for _ in range(300):
        frame()
    for _ in range(max(args.warmup, 1)):
        info = frame()
    assert info_stages is None or info == info_stages, (info, info_stages)
    torch.cuda.synchronize()
    coll.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        info = frame()
    torch.cuda.synchronize()
    coll.barrier()
    dt = max_over_ranks(dist, torch, engine, rehearse, time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({
            "metric": "frames/sec through voxel -> SOR -> RANSAC -> clusters, one frame per GPU",
            "value": world * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": max(args.warmup, 1), "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic" if not rehearse else "synthetic (REHEARSAL: all ranks on one GPU, gloo)",
            "config": {"workload": f"KITTI-like frame of {len(x)} points per GPU through pcrs_obstacle_pipeline "
                                   "(one upload, intermediate clouds on the device), replicas only (no "
                                   "collective)", "name": "c5",
                       "frame_rank0": dict(zip(("voxels", "kept", "ground", "clusters"), info)),
                       "stage_by_stage_ms_rank0": None if dt_stages is None else dt_stages / args.steps * 1e3}}),
            flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=("c2", "c3", "c4", "c5"), default="c2")
    ap.add_argument("--points", type=int, default=0,
                    help="points per GPU (c2), points in all (c4), points per scan (c3); 0 = the config's own size")
    ap.add_argument("--dist-driver", choices=("python", "native"), default="python",
                    help="several GPUs: pointclouds_rs.dist over torch.distributed, or the library's "
                         "own driver over a communicator of its own")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true",
                    help="skip the side timings of the stages around the hot path")
    ap.add_argument("--icp-iters", type=int, default=50, help="ICP iterations for the side metric (0 = skip)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    if args.config == "c3":
        return run_icp(args)
    if args.config == "c5":
        return run_frames(args)
    return run_sor_normals(args)


if __name__ == "__main__":
    sys.exit(main())
