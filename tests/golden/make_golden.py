#!/usr/bin/env python3
"""Generates tests/golden/oracle_vectors.npz with the CPU oracle (oracle/pcrs_oracle.c).

The reference cannot be executed in this environment (Rust, no toolchain), so these vectors pin
the oracle's own behaviour over time and give the GPU tests fixed targets; they are NOT outputs
of the reference.  Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "pointclouds-rs_amd"))

import oracle_ffi  # noqa: E402
from pointclouds_rs import synth  # noqa: E402


def clouds():
    return {
        "uniform64": synth.uniform_cloud(64, 42),
        "uniform1000": synth.uniform_cloud(1000, 42),
        "uniform4096": synth.uniform_cloud(4096, 42),
        "degenerate300": synth.degenerate_cloud(300, 7),
        "hemisphere2000": synth.hemisphere(2000, 99, 5.0),
    }


def main():
    o = oracle_ffi.load()
    out = {}
    for name, (x, y, z) in clouds().items():
        out[f"{name}/xyz"] = np.stack([x, y, z])
        for k in (1, 10, 16, 17):
            idx, dist, d2, cnt = o.knn(x, y, z, x, y, z, k, algo=0)
            out[f"{name}/knn{k}/idx"] = idx
            out[f"{name}/knn{k}/dist"] = dist
            out[f"{name}/knn{k}/count"] = cnt
        for k in (10, 16):
            keep, nk, thr, mean = o.sor(x, y, z, k, 1.0, algo=0)
            out[f"{name}/sor{k}/mean"] = mean
            out[f"{name}/sor{k}/keep"] = keep
            out[f"{name}/sor{k}/thr"] = np.float32(thr)
            out[f"{name}/nrm{k}"] = np.stack(o.normals(x, y, z, k, (0.0, 0.0, 0.0), algo=0))
    # ICP: one accumulation step and a 50-iteration point-to-plane trace on a 2 K hemisphere
    tgt = synth.hemisphere(2000, 99, 5.0)
    tn = o.normals(*tgt, 16, algo=0)
    src = o.apply_transform(*tgt, synth.rot_z(0.05), (0.3, -0.2, 0.1))
    st = o.icp_step(src, tgt, tn, np.inf, algo=0)
    out["icp/src"] = np.stack(src)
    out["icp/tn"] = np.stack(tn)
    out["icp/step/n_corr"] = np.int64(st["n_corr"])
    out["icp/step/plane"] = st["plane"]
    out["icp/step/point"] = st["point"]
    out["icp/step/sum_d2"] = np.float64(st["sum_d2"])
    out["icp/step/tgt_idx"] = st["tgt_idx"]
    # mode 2 = point-to-point with the increment from f64 sums (what the GPU library computes;
    # mode 0 is the reference's own f32 accumulation).  "_tol": with the default tolerance, so
    # that the iteration in which the loop ends and the converged flag are part of the vector.
    for mode, tag in ((0, "p2p"), (1, "p2plane"), (2, "p2p_f64")):
        for tol, suffix in ((0.0, ""), (1e-5, "_tol")):
            rc, res, trace = o.icp(mode, src, tgt, tn, max_iter=50, tol=tol, algo=0, with_trace=True)
            key = f"icp/{tag}{suffix}"
            out[f"{key}/rot"] = np.array(res.rot, dtype=np.float32)
            out[f"{key}/t"] = np.array(res.t, dtype=np.float32)
            out[f"{key}/rmse_fitness_iters"] = np.array([res.rmse, res.fitness, res.num_iterations], dtype=np.float64)
            out[f"{key}/converged"] = np.int32(res.converged)
            out[f"{key}/trace"] = trace
    path = os.path.join(HERE, "oracle_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
    stage_vectors(o)


def flat_clusters(cl):
    """list of index arrays -> (offsets, indices)"""
    off = np.cumsum([0] + [len(c) for c in cl]).astype(np.uint32)
    idx = np.concatenate(cl).astype(np.uint32) if cl else np.zeros(0, np.uint32)
    return off, idx


def stage_vectors(o):
    """tests/golden/stage_vectors.npz: the stages next to the KNN path (radius outlier, voxel
    downsample, plane RANSAC behind its sampling, Euclidean clustering) on three small clouds."""
    from pointclouds_rs import sample_triples

    out = {}
    cl = {
        "uniform3000": synth.uniform_cloud(3000, 5, 0, 20),
        "degenerate300": synth.degenerate_cloud(300, 7),
        "kitti3700": synth.kitti_scene(42, 2000, 500, 200),
    }
    for name, (x, y, z) in cl.items():
        out[f"{name}/xyz"] = np.stack([x, y, z])
        for r, mn in ((0.5, 2), (1.5, 5)):
            keep, _, count = o.radius_outlier(x, y, z, r, mn)
            out[f"{name}/radius{r}_{mn}/keep"] = keep
            out[f"{name}/radius{r}_{mn}/count"] = count
        for size in (0.3, 2.0):
            out[f"{name}/voxel{size}"] = np.stack(o.voxel_downsample(x, y, z, size))
        smp = sample_triples(len(x), 64, 11)
        out[f"{name}/ransac/samples"] = smp
        for thr in (0.05, 0.5):
            nrm, d, inl, best = o.ransac_plane_samples(x, y, z, smp, thr)
            out[f"{name}/ransac{thr}/model"] = np.array([*nrm, d], dtype=np.float32)
            out[f"{name}/ransac{thr}/inliers"] = inl.astype(np.uint32)
            out[f"{name}/ransac{thr}/best"] = np.uint64(best)
        for thr, mn, mx in ((0.4, 1, 10 ** 6), (1.0, 3, 500)):
            off, idx = flat_clusters(o.euclidean_cluster(x, y, z, thr, mn, mx))
            out[f"{name}/cluster{thr}_{mn}_{mx}/offsets"] = off
            out[f"{name}/cluster{thr}_{mn}_{mx}/indices"] = idx
    path = os.path.join(HERE, "stage_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
