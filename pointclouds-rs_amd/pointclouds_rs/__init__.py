"""`pointclouds_rs`-compatible Python surface for the KNN hot path, running on MI355X.

Mirrors the PyO3 module of the reference (crates/python/src/lib.rs:12-49) for the
functions on the accelerated path — same names, argument order, defaults and
error types:

    PointCloud                                  crates/python/src/cloud.rs:4-88
    statistical_outlier_removal(cloud, k, std_mul)      .../filters.rs:36-49
    radius_outlier_removal(cloud, radius, min_neighbors) .../filters.rs:51-64
    estimate_normals(cloud, k)                          .../normals.rs:4-10
    icp_point_to_point / icp_point_to_plane / apply_transform / IcpResult
                                                        .../registration.rs:4-109

and for the stages either side of it:

    voxel_downsample(cloud, voxel_size)                 .../filters.rs:4-13
    ransac_plane(cloud, distance_threshold, iterations) .../segmentation.rs:43-57
    euclidean_cluster(cloud, distance_threshold, min_size, max_size)
                                                        .../segmentation.rs:4-17
    read_pcd / write_pcd                                .../io.rs:4-15

passthrough_filter and the PLY / LAS readers of the reference module are not provided.
Everything computes through libpcrs_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

import math

import numpy as np

from . import _ffi, hip

__all__ = [
    "PointCloud",
    "IcpResult",
    "statistical_outlier_removal",
    "radius_outlier_removal",
    "voxel_downsample",
    "ransac_plane",
    "euclidean_cluster",
    "read_pcd",
    "write_pcd",
    "write_pcd_binary",
    "ransac_plane_seeded",
    "PlaneResult",
    "estimate_normals",
    "icp_point_to_point",
    "icp_point_to_plane",
    "apply_transform",
]


class PointCloud:
    """SoA point cloud (crates/core/src/cloud.rs:3-11) with the PyO3 class's methods."""

    def __init__(self):
        self._x = np.empty(0, dtype=np.float32)
        self._y = np.empty(0, dtype=np.float32)
        self._z = np.empty(0, dtype=np.float32)
        self._normals = None  # (nx, ny, nz) or None, as PointCloud.normals: Option<Normals>
        # (r, g, b) uint8 columns and an f32 column, or None (cloud.rs:8-10: Option<Colors>,
        # Option<Vec<f32>>).  Nothing of the PyO3 surface sets them (only file readers do); they
        # are here so that whatever selects points carries them as cloud.rs:103-140 does.
        self._colors = None
        self._intensity = None

    @classmethod
    def _from_columns(cls, x, y, z, normals=None, colors=None, intensity=None):
        pc = cls()
        pc._x = np.ascontiguousarray(x, dtype=np.float32)
        pc._y = np.ascontiguousarray(y, dtype=np.float32)
        pc._z = np.ascontiguousarray(z, dtype=np.float32)
        pc._normals = normals
        pc._colors = colors
        pc._intensity = intensity
        return pc

    @staticmethod
    def from_numpy(array):
        # crates/python/src/cloud.rs:25-37,91-137
        if not isinstance(array, np.ndarray) or array.dtype not in (np.float32, np.float64):
            raise TypeError("expected NumPy array with dtype float32 or float64, shape (N, 3)")
        if array.ndim != 2:
            raise TypeError("expected NumPy array with dtype float32 or float64, shape (N, 3)")
        if not array.flags["C_CONTIGUOUS"]:
            raise ValueError(
                "array must be C-contiguous (row-major). Use numpy.ascontiguousarray(arr) to convert."
            )
        if array.shape[1] != 3:
            raise ValueError("expected shape (N, 3)")
        a = array.astype(np.float32, copy=False)
        return PointCloud._from_columns(a[:, 0].copy(), a[:, 1].copy(), a[:, 2].copy())

    def len(self):
        return len(self._x)

    def is_empty(self):
        return len(self._x) == 0

    def to_numpy(self):
        return np.stack([self._x, self._y, self._z], axis=1) if len(self._x) else np.empty(
            (0, 3), dtype=np.float32)

    def _check_indices(self, indices):
        idx = np.asarray(list(indices), dtype=np.int64)
        if idx.size and (idx.min() < 0 or idx.max() >= len(self._x)):
            bad = int(idx[(idx < 0) | (idx >= len(self._x))][0])
            raise IndexError(f"index {bad} out of bounds for cloud with {len(self._x)} points")
        return idx

    def select(self, indices):
        # crates/core/src/cloud.rs:103-140 (normals, colors and intensity travel with the points)
        idx = self._check_indices(indices)
        normals = colors = intensity = None
        if self._normals is not None:
            normals = tuple(c[idx] for c in self._normals)
        if self._colors is not None:
            colors = tuple(c[idx] for c in self._colors)
        if self._intensity is not None:
            intensity = self._intensity[idx]
        return PointCloud._from_columns(self._x[idx], self._y[idx], self._z[idx], normals, colors,
                                        intensity)

    def select_inverse(self, indices):
        # crates/core/src/cloud.rs:152-162
        idx = self._check_indices(indices)
        keep = np.ones(len(self._x), dtype=bool)
        keep[idx] = False
        return self.select(np.nonzero(keep)[0])

    def __len__(self):
        return len(self._x)

    def __repr__(self):
        return f"PointCloud(n={len(self._x)})"

    # Extension (the reference class has no getter for normals): (N, 3) array or None.
    def normals_numpy(self):
        if self._normals is None:
            return None
        return np.stack(self._normals, axis=1)


class IcpResult:
    """crates/python/src/registration.rs:4-29"""

    def __init__(self, converged, fitness, rmse, num_iterations, translation, rotation):
        self.converged = bool(converged)
        self.fitness = float(fitness)
        self.rmse = float(rmse)
        self.num_iterations = int(num_iterations)
        self.translation = [float(v) for v in translation]
        self.rotation = [[float(rotation[r * 3 + c]) for c in range(3)] for r in range(3)]

    def __repr__(self):
        return (f"IcpResult(converged={'true' if self.converged else 'false'}, "
                f"rmse={self.rmse:.6f}, iterations={self.num_iterations})")


def _require_cloud(c, name="cloud"):
    if not isinstance(c, PointCloud):
        raise TypeError(f"argument '{name}': expected PointCloud")


def _f32(v):
    """A Python float as the PyO3 signature receives it: converted to f32 before the function
    body runs, so a value beyond the f32 range is already infinite when it is checked."""
    with np.errstate(over="ignore"):
        return float(np.float32(float(v)))


def _usize(v):
    """PyO3's usize extraction: negative values fail at argument conversion, before any check
    in the function body."""
    v = int(v)
    if v < 0:
        raise OverflowError("can't convert negative int to unsigned")
    return v


def statistical_outlier_removal(cloud, k, std_mul):
    """crates/python/src/filters.rs:36-49 -> crates/filters/src/statistical_outlier.rs:4-69"""
    _require_cloud(cloud)
    k = _usize(k)
    std_mul = _f32(std_mul)
    if not math.isfinite(std_mul) or std_mul < 0.0:
        raise ValueError("std_mul must be >= 0 and finite")
    keep, _, _, _ = hip.sor(cloud._x, cloud._y, cloud._z, k, std_mul)
    return cloud.select(np.nonzero(keep)[0])


def voxel_downsample(cloud, voxel_size):
    """crates/python/src/filters.rs:4-13 -> crates/filters/src/voxel_downsample.rs:12-65"""
    _require_cloud(cloud)
    voxel_size = _f32(voxel_size)
    if not math.isfinite(voxel_size) or voxel_size <= 0.0:
        raise ValueError("voxel_size must be > 0 and finite")
    if cloud.is_empty():
        return PointCloud()
    x, y, z = hip.voxel_downsample(cloud._x, cloud._y, cloud._z, voxel_size)
    return PointCloud._from_columns(x, y, z, None)


def read_pcd(path):
    """crates/python/src/io.rs:4-9 -> crates/io/src/pcd.rs:7-21; IOError on a bad file."""
    from . import io as _io

    try:
        x, y, z = _io.read_pcd_columns(path)
    except ValueError as e:
        raise IOError(str(e))
    return PointCloud._from_columns(x, y, z, None)


def write_pcd(path, cloud):
    """crates/python/src/io.rs:11-15 -> crates/io/src/pcd.rs:24-43 (ASCII)."""
    from . import io as _io

    _require_cloud(cloud)
    _io.write_pcd_columns(path, cloud._x, cloud._y, cloud._z)


def write_pcd_binary(path, cloud):
    """crates/io/src/pcd.rs:46-71."""
    from . import io as _io

    _require_cloud(cloud)
    _io.write_pcd_binary_columns(path, cloud._x, cloud._y, cloud._z)


def euclidean_cluster(cloud, distance_threshold, min_size, max_size):
    """crates/python/src/segmentation.rs:4-17 -> euclidean_cluster.rs:96-187: list of index lists."""
    _require_cloud(cloud)
    if min_size < 0 or max_size < 0:
        raise OverflowError("can't convert negative int to unsigned")
    max_size = min(int(max_size), (1 << 64) - 1)
    out = hip.euclidean_cluster(cloud._x, cloud._y, cloud._z, float(distance_threshold),
                                int(min_size), max_size)
    return [[int(i) for i in c] for c in out]


class PlaneResult:
    """crates/python/src/segmentation.rs:19-41"""

    def __init__(self, normal, d, inliers):
        self.normal = [float(v) for v in normal]
        self.d = float(d)
        self.inliers = [int(i) for i in inliers]

    def __repr__(self):
        return "PlaneResult(normal=%r, d=%.4f, inliers=%d)" % (self.normal, self.d, len(self.inliers))


def _sample_triples_scalar(n, iterations, seed):
    """sample_triples, draw by draw (the definition; the vectorised road must equal it)."""
    mask = (1 << 64) - 1
    state = [int(seed) & mask]

    def below(m):
        state[0] = (state[0] + 0x9E3779B97F4A7C15) & mask
        v = state[0]
        v = ((v ^ (v >> 30)) * 0xBF58476D1CE4E5B9) & mask
        v = ((v ^ (v >> 27)) * 0x94D049BB133111EB) & mask
        v ^= v >> 31
        return (v * m) >> 64

    out = []
    for _ in range(int(iterations)):
        i0 = below(n)
        i1 = below(n)
        attempts = 0
        while i1 == i0 and attempts <= 100:
            i1 = below(n)
            attempts += 1
        if i1 == i0:
            continue
        i2 = below(n)
        attempts = 0
        while (i2 == i0 or i2 == i1) and attempts <= 100:
            i2 = below(n)
            attempts += 1
        if i2 == i0 or i2 == i1:
            continue
        out.append((i0, i1, i2))
    return np.array(out, dtype=np.uint32).reshape(-1, 3)


def sample_triples(n, iterations, seed):
    """The sampling procedure of ransac_plane.rs:141-165 (three distinct indices, up to 100 retries
    each) over a SplitMix64 stream.  The reference draws from rand's StdRng (ChaCha12), which
    cannot be reproduced without that crate: a Rust caller passes its own triples to
    pcrs_ransac_plane_samples, this generator only serves the Python mirror.

    SplitMix64 is a function of the draw counter, so all draws of a call are computed at once;
    when no triple needs a redraw (nearly always, on clouds of thousands of points) the triples
    are those draws three by three, otherwise the call takes the draw-by-draw road."""
    iterations = int(iterations)
    if n < 3:
        return np.zeros((0, 3), dtype=np.uint32)
    if iterations <= 0 or n >= (1 << 32):
        return _sample_triples_scalar(n, iterations, seed)
    with np.errstate(over="ignore"):
        k = np.arange(1, 3 * iterations + 1, dtype=np.uint64)
        v = np.uint64(int(seed) & ((1 << 64) - 1)) + k * np.uint64(0x9E3779B97F4A7C15)
        v = (v ^ (v >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        v = (v ^ (v >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        v ^= v >> np.uint64(31)
        # (v * n) >> 64 for n < 2^32, in 64-bit pieces
        m = np.uint64(n)
        hi = (v >> np.uint64(32)) * m
        lo = ((v & np.uint64(0xFFFFFFFF)) * m) >> np.uint64(32)
        t = ((hi + lo) >> np.uint64(32)).reshape(iterations, 3)
    if ((t[:, 0] != t[:, 1]) & (t[:, 2] != t[:, 0]) & (t[:, 2] != t[:, 1])).all():
        return t.astype(np.uint32)
    return _sample_triples_scalar(n, iterations, seed)


def ransac_plane_seeded(cloud, distance_threshold, iterations, seed):
    """crates/segmentation/src/ransac_plane.rs:56-138 with the mirror's own sample stream."""
    _require_cloud(cloud)
    if iterations < 0:
        raise OverflowError("can't convert negative int to unsigned")
    samples = sample_triples(cloud.len(), iterations, seed)
    normal, d, inliers, _ = hip.ransac_plane_samples(cloud._x, cloud._y, cloud._z, samples,
                                                     float(distance_threshold))
    return PlaneResult(normal, d, inliers)


def ransac_plane(cloud, distance_threshold, iterations):
    """crates/python/src/segmentation.rs:43-57: random seed (ransac_plane.rs:36-43)."""
    import secrets

    return ransac_plane_seeded(cloud, distance_threshold, iterations, secrets.randbits(64))


def radius_outlier_removal(cloud, radius, min_neighbors):
    """crates/python/src/filters.rs:51-64 -> crates/filters/src/radius_outlier.rs:4-18"""
    _require_cloud(cloud)
    radius = _f32(radius)
    min_neighbors = _usize(min_neighbors)
    if not math.isfinite(radius) or radius <= 0.0:
        raise ValueError("radius must be > 0 and finite")
    if cloud.is_empty():
        return PointCloud()
    keep, _, _ = hip.radius_outlier(cloud._x, cloud._y, cloud._z, radius, min_neighbors)
    return cloud.select(np.nonzero(keep)[0])


def estimate_normals(cloud, k):
    """crates/python/src/normals.rs:4-10: returns a copy of the cloud with normals attached."""
    _require_cloud(cloud)
    if k < 0:
        raise OverflowError("can't convert negative int to unsigned")
    nx, ny, nz = hip.normals(cloud._x, cloud._y, cloud._z, int(k), (0.0, 0.0, 0.0))
    return PointCloud._from_columns(
        cloud._x.copy(), cloud._y.copy(), cloud._z.copy(), (nx, ny, nz),
        None if cloud._colors is None else tuple(c.copy() for c in cloud._colors),
        None if cloud._intensity is None else cloud._intensity.copy())


def _icp(mode, source, target, normals, max_iterations, tolerance, max_correspondence_distance):
    res = hip.icp_register(
        mode,
        (source._x, source._y, source._z),
        (target._x, target._y, target._z),
        normals,
        int(max_iterations),
        float(tolerance),
        float(max_correspondence_distance),
    )
    return IcpResult(res.converged, res.fitness, res.rmse, res.num_iterations,
                     list(res.translation), list(res.rotation))


def icp_point_to_point(source, target, max_iterations=50, tolerance=1e-5,
                       max_correspondence_distance=math.inf):
    """crates/python/src/registration.rs:31-55"""
    _require_cloud(source, "source")
    _require_cloud(target, "target")
    return _icp(_ffi.ICP_POINT_TO_POINT, source, target, None, max_iterations, tolerance,
                max_correspondence_distance)


def icp_point_to_plane(source, target, max_iterations=50, tolerance=1e-5,
                       max_correspondence_distance=math.inf):
    """crates/python/src/registration.rs:57-95"""
    _require_cloud(source, "source")
    _require_cloud(target, "target")
    if target._normals is None:
        raise ValueError("target cloud must have normals for point-to-plane ICP. "
                         "Use estimate_normals(target, k) first.")
    try:
        return _icp(_ffi.ICP_POINT_TO_PLANE, source, target, target._normals, max_iterations,
                    tolerance, max_correspondence_distance)
    except _ffi.PcrsError as e:
        if e.code == _ffi.PCRS_ERR_NORMALS_MISMATCH:
            raise ValueError(e.message) from None
        raise


def apply_transform(cloud, rotation, translation):
    """crates/python/src/registration.rs:97-109 (drops normals, icp.rs:91)"""
    _require_cloud(cloud)
    r = np.asarray(rotation, dtype=np.float32)
    t = np.asarray(translation, dtype=np.float32)
    if r.shape != (3, 3) or t.shape != (3,):
        raise TypeError("rotation must be 3x3 and translation length 3")
    if cloud.is_empty():
        return PointCloud()
    x, y, z = hip.apply_transform(cloud._x, cloud._y, cloud._z, r, t)
    return PointCloud._from_columns(x, y, z)
