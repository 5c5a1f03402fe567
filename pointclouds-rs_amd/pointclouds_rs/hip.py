"""NumPy-level access to the C-ABI (host-pointer entry points).

Thin wrappers: contiguous float32 columns in, NumPy arrays out.  Every function
goes through libpcrs_hip.so; nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import sys

import numpy as np

from . import _ffi
from ._ffi import get_context

_V = C.c_void_p


def _col(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 1:
        raise ValueError("expected a 1-D column")
    return a


def _ptr(a):
    return a.ctypes.data_as(_V) if a is not None and a.size else _V(None)


def _vec3(v):
    return (C.c_float * 3)(*[float(t) for t in v])


def _mat9(r):
    r = np.asarray(r, dtype=np.float32).reshape(9)
    return (C.c_float * 9)(*[float(t) for t in r])


class Index:
    """Device-resident search structure over one cloud (KdTree::build, kdtree.rs:25-44)."""

    def __init__(self, x, y, z, k_hint=0, algo=_ffi.ALGO_AUTO, device=0):
        self.ctx = get_context(device)
        self._cols = (_col(x), _col(y), _col(z))
        if not (len(self._cols[0]) == len(self._cols[1]) == len(self._cols[2])):
            raise ValueError("x, y, z must have the same length")
        h = _V()
        self.ctx.check(
            self.ctx.lib.pcrs_index_build(
                self.ctx.handle, *[_ptr(c) for c in self._cols], len(self._cols[0]), k_hint, algo,
                C.byref(h),
            )
        )
        self.handle = h

    def __len__(self):
        return int(self.ctx.lib.pcrs_index_len(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.lib.pcrs_index_destroy(self.handle)
            self.handle = None

    def __del__(self):
        # at interpreter shutdown the HIP runtime may already be gone: the process frees it all
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def knn(self, qx, qy, qz, k):
        """Batched KdTree::knn: (idx[n,k] uint32, dist[n,k] float32, count[n] uint32)."""
        qx, qy, qz = _col(qx), _col(qy), _col(qz)
        n = len(qx)
        idx = np.empty((n, k), dtype=np.uint32)
        dist = np.empty((n, k), dtype=np.float32)
        count = np.zeros(n, dtype=np.uint32)
        self.ctx.check(
            self.ctx.lib.pcrs_knn_batch(self.handle, _ptr(qx), _ptr(qy), _ptr(qz), n, k, _ptr(idx),
                                        _ptr(dist), _ptr(count))
        )
        return idx, dist, count

    def radius_count(self, qx, qy, qz, radius):
        qx, qy, qz = _col(qx), _col(qy), _col(qz)
        n = len(qx)
        count = np.zeros(n, dtype=np.uint32)
        self.ctx.check(
            self.ctx.lib.pcrs_radius_count(self.handle, _ptr(qx), _ptr(qy), _ptr(qz), n,
                                           float(radius), _ptr(count))
        )
        return count

    def radius_search(self, qx, qy, qz, radius):
        """Batched KdTree::radius_search (kdtree.rs:105-135) as CSR: (offsets[n+1], indices);
        indices[offsets[i]:offsets[i+1]] are the points within `radius` of query i, ascending."""
        qx, qy, qz = _col(qx), _col(qy), _col(qz)
        n = len(qx)
        offsets = np.zeros(n + 1, dtype=np.uint32)
        total = C.c_size_t(0)
        lib = self.ctx.lib
        # first call sizes the list (it reports the total and returns INVALID_ARG when there
        # are results and no room for them)
        rc = lib.pcrs_radius_search(self.handle, _ptr(qx), _ptr(qy), _ptr(qz), n, float(radius),
                                    _ptr(offsets), None, 0, C.byref(total))
        if total.value == 0 or total.value >= 0xFFFFFFFF:
            self.ctx.check(rc)
            return offsets, np.empty(0, dtype=np.uint32)
        indices = np.empty(total.value, dtype=np.uint32)
        self.ctx.check(lib.pcrs_radius_search(self.handle, _ptr(qx), _ptr(qy), _ptr(qz), n,
                                              float(radius), _ptr(offsets), _ptr(indices),
                                              total.value, C.byref(total)))
        return offsets, indices

    def find_correspondences(self, sx, sy, sz, max_distance=np.inf):
        """Dense find_correspondences: (tgt_idx[n] uint32 (NO_INDEX = none), dist[n], n_corr)."""
        sx, sy, sz = _col(sx), _col(sy), _col(sz)
        n = len(sx)
        tgt = np.empty(n, dtype=np.uint32)
        dist = np.empty(n, dtype=np.float32)
        nc = C.c_size_t(0)
        self.ctx.check(
            self.ctx.lib.pcrs_find_correspondences(self.handle, _ptr(sx), _ptr(sy), _ptr(sz), n,
                                                   float(max_distance), _ptr(tgt), _ptr(dist),
                                                   C.byref(nc))
        )
        return tgt, dist, int(nc.value)

    def last_stats(self):
        s = (C.c_uint64 * 3)()
        self.ctx.check(self.ctx.lib.pcrs_index_last_stats(self.handle, s))
        return {"grid_tile": int(s[0]), "grid_ring": int(s[1]), "brute": int(s[2])}

    def levels(self):
        """(finer, coarser): the grids besides its own that the index holds for uneven clouds."""
        a, b = C.c_uint32(0), C.c_uint32(0)
        self.ctx.check(self.ctx.lib.pcrs_index_levels(self.handle, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)


def sor(x, y, z, k, std_mul, device=0):
    """statistical_outlier_removal on columns: (keep[n] bool, n_keep, threshold, mean[n])."""
    ctx = get_context(device)
    x, y, z = _col(x), _col(y), _col(z)
    n = len(x)
    keep = np.zeros(n, dtype=np.uint8)
    mean = np.full(n, np.nan, dtype=np.float32)
    n_keep = C.c_size_t(0)
    thr = C.c_float(np.nan)
    ctx.check(
        ctx.lib.pcrs_sor(ctx.handle, _ptr(x), _ptr(y), _ptr(z), n, k, float(std_mul), _ptr(keep),
                         C.byref(n_keep), C.byref(thr), _ptr(mean))
    )
    return keep.astype(bool), int(n_keep.value), float(thr.value), mean


def radius_outlier(x, y, z, radius, min_neighbors, device=0):
    """radius_outlier_removal on columns: (keep[n] bool, n_keep, count[n])."""
    ctx = get_context(device)
    x, y, z = _col(x), _col(y), _col(z)
    n = len(x)
    keep = np.zeros(n, dtype=np.uint8)
    count = np.zeros(n, dtype=np.uint32)
    n_keep = C.c_size_t(0)
    ctx.check(
        ctx.lib.pcrs_radius_outlier(ctx.handle, _ptr(x), _ptr(y), _ptr(z), n, float(radius),
                                    int(min_neighbors), _ptr(keep), _ptr(count), C.byref(n_keep))
    )
    return keep.astype(bool), int(n_keep.value), count


def voxel_downsample(x, y, z, voxel_size, device=0):
    """voxel_downsample on columns: centroid columns (x, y, z) in voxel-key order."""
    ctx = get_context(device)
    x, y, z = _col(x), _col(y), _col(z)
    n = len(x)
    out = [np.empty(max(n, 1), dtype=np.float32) for _ in range(3)]
    n_out = C.c_size_t(0)
    ctx.check(
        ctx.lib.pcrs_voxel_downsample(ctx.handle, _ptr(x), _ptr(y), _ptr(z), n, float(voxel_size),
                                      *[_ptr(o) for o in out], C.byref(n_out))
    )
    return tuple(o[: n_out.value].copy() for o in out)


def ransac_plane_samples(x, y, z, samples, distance_threshold, device=0):
    """ransac_plane_seeded behind its sample generation: samples is an (S, 3) index array.
    Returns (normal[3], d, inlier indices ascending, best inlier count)."""
    ctx = get_context(device)
    x, y, z = _col(x), _col(y), _col(z)
    n = len(x)
    smp = np.ascontiguousarray(np.asarray(samples, dtype=np.uint32).reshape(-1, 3))
    model = (C.c_float * 4)()
    mask = np.zeros(max(n, 1), dtype=np.uint8)
    n_in = C.c_size_t(0)
    best = C.c_uint64(0)
    ctx.check(
        ctx.lib.pcrs_ransac_plane_samples(ctx.handle, _ptr(x), _ptr(y), _ptr(z), n, _ptr(smp),
                                          len(smp), float(distance_threshold), model, _ptr(mask),
                                          C.byref(n_in), C.byref(best))
    )
    inliers = np.nonzero(mask[:n])[0]
    assert len(inliers) == n_in.value
    return np.array(model[:3], dtype=np.float32), float(model[3]), inliers, int(best.value)


def euclidean_cluster(x, y, z, distance_threshold, min_size, max_size, device=0):
    """euclidean_cluster on columns: list of index arrays, largest cluster first."""
    ctx = get_context(device)
    x, y, z = _col(x), _col(y), _col(z)
    n = len(x)
    offsets = np.zeros(n + 1, dtype=np.uint32)
    indices = np.zeros(max(n, 1), dtype=np.uint32)
    nc = C.c_size_t(0)
    ctx.check(
        ctx.lib.pcrs_euclidean_cluster(ctx.handle, _ptr(x), _ptr(y), _ptr(z), n,
                                       float(distance_threshold), int(min_size), int(max_size),
                                       _ptr(offsets), _ptr(indices), C.byref(nc))
    )
    return [indices[offsets[c]:offsets[c + 1]].copy() for c in range(nc.value)]


def sample_triples(n, iterations, seed):
    """pcrs_sample_triples: the (S, 3) index triples of the mirror's SplitMix64 stream."""
    lib = _ffi.load_library()
    out = np.zeros((max(int(iterations), 1), 3), dtype=np.uint32)
    ns = C.c_size_t(0)
    rc = lib.pcrs_sample_triples(int(n), int(iterations), int(seed) & ((1 << 64) - 1), _ptr(out),
                                 C.byref(ns))
    if rc != 0:
        raise _ffi.PcrsError(rc, "pcrs_sample_triples")
    return out[: ns.value].copy()


def obstacle_pipeline(x, y, z, voxel_size=0.15, sor_k=20, sor_std_mul=2.0, plane_threshold=0.15,
                      plane_iterations=500, plane_seed=1234, cluster_threshold=0.8, cluster_min_size=10,
                      cluster_max_size=20000, want_obstacles=True, device=0):
    """voxel_downsample -> statistical_outlier_removal -> ransac_plane -> drop the plane ->
    euclidean_cluster in one call (pcrs_obstacle_pipeline).  Returns a dict: the stage counts,
    the plane (normal, d), the SOR threshold, the obstacle cloud (x, y, z) and the clusters as a
    list of index arrays into it, largest first."""
    ctx = get_context(device)
    x, y, z = _col(x), _col(y), _col(z)
    n = len(x)
    prm = _ffi.ObstacleParamsC(float(voxel_size), int(sor_k), float(sor_std_mul), float(plane_threshold),
                               int(plane_iterations), int(plane_seed) & ((1 << 64) - 1),
                               float(cluster_threshold), int(cluster_min_size), int(cluster_max_size))
    res = _ffi.ObstacleResultC()
    obst = [np.empty(max(n, 1), dtype=np.float32) for _ in range(3)] if want_obstacles else [None] * 3
    offsets = np.zeros(n + 1, dtype=np.uint32)
    indices = np.zeros(max(n, 1), dtype=np.uint32)
    ctx.check(ctx.lib.pcrs_obstacle_pipeline(ctx.handle, _ptr(x), _ptr(y), _ptr(z), n, C.byref(prm),
                                             C.byref(res), *[_ptr(o) for o in obst], _ptr(offsets),
                                             _ptr(indices)))
    no = int(res.n_obstacles)
    return dict(n_voxels=int(res.n_voxels), n_kept=int(res.n_kept), n_ground=int(res.n_ground),
                n_obstacles=no, plane_normal=np.array(res.plane[:3], dtype=np.float32),
                plane_d=float(res.plane[3]), sor_threshold=float(res.sor_threshold),
                obstacles=tuple(o[:no] for o in obst) if want_obstacles else None,
                clusters=[indices[offsets[c]:offsets[c + 1]] for c in range(int(res.n_clusters))])


def normals(x, y, z, k, viewpoint=(0.0, 0.0, 0.0), device=0):
    """estimate_normals_with_viewpoint on columns: (nx, ny, nz); empty when k == 0."""
    ctx = get_context(device)
    x, y, z = _col(x), _col(y), _col(z)
    n = len(x)
    if n == 0 or k == 0:
        e = np.empty(0, dtype=np.float32)
        return e, e.copy(), e.copy()
    out = [np.empty(n, dtype=np.float32) for _ in range(3)]
    ctx.check(
        ctx.lib.pcrs_normals(ctx.handle, _ptr(x), _ptr(y), _ptr(z), n, k, _vec3(viewpoint),
                             *[_ptr(o) for o in out])
    )
    return tuple(out)


def sor_normals(x, y, z, k_sor, std_mul, k_normals, viewpoint=(0.0, 0.0, 0.0), device=0):
    """One upload + one neighbour search for both consumers."""
    ctx = get_context(device)
    x, y, z = _col(x), _col(y), _col(z)
    n = len(x)
    keep = np.zeros(n, dtype=np.uint8)
    out = [np.empty(n, dtype=np.float32) for _ in range(3)]
    n_keep = C.c_size_t(0)
    thr = C.c_float(np.nan)
    ctx.check(
        ctx.lib.pcrs_sor_normals(ctx.handle, _ptr(x), _ptr(y), _ptr(z), n, k_sor, float(std_mul),
                                 k_normals, _vec3(viewpoint), _ptr(keep), C.byref(n_keep),
                                 C.byref(thr), *[_ptr(o) for o in out])
    )
    return keep.astype(bool), int(n_keep.value), float(thr.value), tuple(out)


def apply_transform(x, y, z, rotation, translation, device=0):
    ctx = get_context(device)
    x, y, z = _col(x), _col(y), _col(z)
    n = len(x)
    out = [np.empty(n, dtype=np.float32) for _ in range(3)]
    ctx.check(
        ctx.lib.pcrs_apply_transform(ctx.handle, _ptr(x), _ptr(y), _ptr(z), n, _mat9(rotation),
                                     _vec3(translation), *[_ptr(o) for o in out])
    )
    return tuple(out)


def compose(ra, ta, rb, tb):
    lib = _ffi.load_library()
    ro = (C.c_float * 9)()
    to = (C.c_float * 3)()
    lib.pcrs_compose(_mat9(ra), _vec3(ta), _mat9(rb), _vec3(tb), ro, to)
    return np.array(ro, dtype=np.float32).reshape(3, 3), np.array(to, dtype=np.float32)


class Icp:
    """Device-resident ICP state (target index, target normals, working source copy)."""

    def __init__(self, source, target, target_normals=None, device=0):
        self.ctx = get_context(device)
        s = [_col(c) for c in source]
        t = [_col(c) for c in target]
        nn = [_col(c) for c in target_normals] if target_normals is not None else [None] * 3
        n_normals = len(nn[0]) if nn[0] is not None else 0
        h = _V()
        self.ctx.check(
            self.ctx.lib.pcrs_icp_create(self.ctx.handle, *[_ptr(c) for c in s], len(s[0]),
                                         *[_ptr(c) for c in t], len(t[0]),
                                         *[_ptr(c) for c in nn], n_normals, C.byref(h))
        )
        self.handle = h
        self.n = len(s[0])

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.lib.pcrs_icp_destroy(self.handle)
            self.handle = None

    def __del__(self):
        # at interpreter shutdown the HIP runtime may already be gone: the process frees it all
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.ctx.check(self.ctx.lib.pcrs_icp_reset(self.handle))

    def accumulate(self, mode, max_distance=np.inf):
        sums = np.zeros(_ffi.ICP_NSUMS, dtype=np.float64)
        self.ctx.check(
            self.ctx.lib.pcrs_icp_accumulate(self.handle, mode, float(max_distance),
                                             sums.ctypes.data_as(_ffi.c_f64_p))
        )
        return sums

    def apply(self, rotation, translation):
        self.ctx.check(self.ctx.lib.pcrs_icp_apply(self.handle, _mat9(rotation), _vec3(translation)))

    def current(self):
        out = [np.empty(self.n, dtype=np.float32) for _ in range(3)]
        self.ctx.check(self.ctx.lib.pcrs_icp_get_current(self.handle, *[_ptr(o) for o in out]))
        return tuple(out)

    def run(self, mode, max_iterations=50, tolerance=1e-5, max_correspondence_distance=np.inf):
        res = _ffi.IcpResultC()
        self.ctx.check(
            self.ctx.lib.pcrs_icp_run(self.handle, mode, max_iterations, float(tolerance),
                                      float(max_correspondence_distance), C.byref(res))
        )
        return res


def icp_solve(mode, sums):
    lib = _ffi.load_library()
    sums = np.ascontiguousarray(sums, dtype=np.float64)
    rot = (C.c_float * 9)()
    t = (C.c_float * 3)()
    rc = lib.pcrs_icp_solve(mode, sums.ctypes.data_as(_ffi.c_f64_p), rot, t)
    if rc != 0:
        raise _ffi.PcrsError(rc, "pcrs_icp_solve")
    return np.array(rot, dtype=np.float32).reshape(3, 3), np.array(t, dtype=np.float32)


def icp_register(mode, source, target, target_normals=None, max_iterations=50, tolerance=1e-5,
                 max_correspondence_distance=np.inf, device=0):
    """Whole registration on host columns (pcrs_icp_register)."""
    ctx = get_context(device)
    s = [_col(c) for c in source]
    t = [_col(c) for c in target]
    nn = [_col(c) for c in target_normals] if target_normals is not None else [None] * 3
    n_normals = len(nn[0]) if nn[0] is not None else 0
    res = _ffi.IcpResultC()
    ctx.check(
        ctx.lib.pcrs_icp_register(ctx.handle, mode, *[_ptr(c) for c in s], len(s[0]),
                                  *[_ptr(c) for c in t], len(t[0]), *[_ptr(c) for c in nn],
                                  n_normals, max_iterations, float(tolerance),
                                  float(max_correspondence_distance), C.byref(res))
    )
    return res
