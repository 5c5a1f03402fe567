"""ctypes access to the CPU oracle (oracle/libpcrs_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libpcrs_oracle.so")
NO_INDEX = 0xFFFFFFFF
_V = C.c_void_p


class IcpResult(C.Structure):
    _fields_ = [("rot", C.c_float * 9), ("t", C.c_float * 3), ("fitness", C.c_float),
                ("rmse", C.c_float), ("converged", C.c_int32), ("num_iterations", C.c_uint32)]


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(_V) if a is not None and a.size else _V(None)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        L = lib
        L.orc_max_threads.restype = C.c_int
        L.orc_knn_batch.restype = None
        L.orc_knn_batch.argtypes = [_V, _V, _V, C.c_size_t, _V, _V, _V, C.c_size_t, C.c_uint32,
                                    C.c_int, C.c_int, _V, _V, _V, _V]
        L.orc_sor_mean_dist.restype = None
        L.orc_sor_mean_dist.argtypes = [_V, _V, _V, C.c_size_t, C.c_uint32, C.c_int, C.c_int, _V]
        L.orc_sor_threshold.restype = C.c_int
        L.orc_sor_threshold.argtypes = [_V, C.c_size_t, C.c_float, C.POINTER(C.c_float),
                                        C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_sor.restype = C.c_size_t
        L.orc_sor.argtypes = [_V, _V, _V, C.c_size_t, C.c_uint32, C.c_float, C.c_int, C.c_int, _V,
                              C.POINTER(C.c_float), _V]
        L.orc_radius_outlier.restype = C.c_size_t
        L.orc_radius_outlier.argtypes = [_V, _V, _V, C.c_size_t, C.c_float, C.c_size_t, C.c_int,
                                         _V, _V]
        L.orc_smallest_eigenvector_3x3.restype = None
        L.orc_smallest_eigenvector_3x3.argtypes = [C.c_float] * 6 + [C.POINTER(C.c_float)]
        L.orc_normals.restype = None
        L.orc_normals.argtypes = [_V, _V, _V, C.c_size_t, C.c_uint32, C.POINTER(C.c_float),
                                  C.c_int, C.c_int, _V, _V, _V]
        L.orc_apply_transform.restype = None
        L.orc_apply_transform.argtypes = [_V, _V, _V, C.c_size_t, C.POINTER(C.c_float),
                                          C.POINTER(C.c_float), _V, _V, _V]
        L.orc_compose.restype = None
        L.orc_compose.argtypes = [C.POINTER(C.c_float)] * 6
        L.orc_find_correspondences.restype = C.c_size_t
        L.orc_find_correspondences.argtypes = [_V, _V, _V, C.c_size_t, _V, _V, _V, C.c_size_t,
                                               C.c_float, C.c_int, C.c_int, _V, _V]
        L.orc_solve_plane_system.restype = None
        L.orc_solve_plane_system.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double),
                                             C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_icp_step.restype = C.c_size_t
        L.orc_icp_step.argtypes = [_V, _V, _V, C.c_size_t, _V, _V, _V, _V, _V, _V, C.c_size_t,
                                   C.c_float, C.c_int, C.c_int, _V, _V, C.POINTER(C.c_float),
                                   C.POINTER(C.c_double), _V, _V]
        L.orc_icp.restype = C.c_int
        L.orc_icp.argtypes = [C.c_int, _V, _V, _V, C.c_size_t, _V, _V, _V, _V, _V, _V, C.c_size_t,
                              C.c_size_t, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_int,
                              C.POINTER(IcpResult), _V]

    def max_threads(self):
        return int(self.lib.orc_max_threads())

    def knn(self, rx, ry, rz, qx, qy, qz, k, algo=0, threads=0):
        rx, ry, rz, qx, qy, qz = map(_f, (rx, ry, rz, qx, qy, qz))
        n, m = len(qx), len(rx)
        idx = np.full((n, k), NO_INDEX, dtype=np.uint32)
        dist = np.full((n, k), np.nan, dtype=np.float32)
        d2 = np.full((n, k), np.nan, dtype=np.float32)
        count = np.zeros(n, dtype=np.uint32)
        self.lib.orc_knn_batch(_p(rx), _p(ry), _p(rz), m, _p(qx), _p(qy), _p(qz), n, k, algo,
                               threads, _p(idx), _p(dist), _p(d2), _p(count))
        return idx, dist, d2, count

    def sor_mean_dist(self, x, y, z, k, algo=0, threads=0):
        x, y, z = map(_f, (x, y, z))
        mean = np.empty(len(x), dtype=np.float32)
        self.lib.orc_sor_mean_dist(_p(x), _p(y), _p(z), len(x), k, algo, threads, _p(mean))
        return mean

    def sor_threshold(self, mean, std_mul):
        mean = _f(mean)
        thr, mu, sd = C.c_float(), C.c_float(), C.c_float()
        ok = self.lib.orc_sor_threshold(_p(mean), len(mean), std_mul, C.byref(thr), C.byref(mu),
                                        C.byref(sd))
        return bool(ok), thr.value, mu.value, sd.value

    def sor(self, x, y, z, k, std_mul, algo=0, threads=0):
        x, y, z = map(_f, (x, y, z))
        n = len(x)
        keep = np.zeros(n, dtype=np.uint8)
        mean = np.full(n, np.nan, dtype=np.float32)
        thr = C.c_float(np.nan)
        kept = self.lib.orc_sor(_p(x), _p(y), _p(z), n, k, std_mul, algo, threads, _p(keep),
                                C.byref(thr), _p(mean))
        return keep.astype(bool), int(kept), thr.value, mean

    def radius_outlier(self, x, y, z, radius, min_neighbors, threads=0):
        x, y, z = map(_f, (x, y, z))
        n = len(x)
        keep = np.zeros(n, dtype=np.uint8)
        count = np.zeros(n, dtype=np.uint32)
        kept = self.lib.orc_radius_outlier(_p(x), _p(y), _p(z), n, radius, min_neighbors, threads,
                                           _p(keep), _p(count))
        return keep.astype(bool), int(kept), count

    def radius_search(self, x, y, z, qx, qy, qz, radius):
        """(offsets[n+1], indices): ascending indices of the points within `radius` per query."""
        x, y, z, qx, qy, qz = map(_f, (x, y, z, qx, qy, qz))
        m, n = len(x), len(qx)
        self.lib.orc_radius_search.restype = C.c_size_t
        self.lib.orc_radius_search.argtypes = [_V, _V, _V, C.c_size_t, _V, _V, _V, C.c_size_t,
                                               C.c_float, _V, _V, C.c_size_t]
        offsets = np.zeros(n + 1, dtype=np.uint32)
        total = self.lib.orc_radius_search(_p(x), _p(y), _p(z), m, _p(qx), _p(qy), _p(qz), n,
                                           radius, _p(offsets), None, 0)
        indices = np.zeros(max(total, 1), dtype=np.uint32)
        self.lib.orc_radius_search(_p(x), _p(y), _p(z), m, _p(qx), _p(qy), _p(qz), n, radius,
                                   _p(offsets), _p(indices), total)
        return offsets, indices[:total]

    def voxel_downsample(self, x, y, z, voxel_size):
        x, y, z = map(_f, (x, y, z))
        n = len(x)
        out = [np.zeros(max(n, 1), dtype=np.float32) for _ in range(3)]
        self.lib.orc_voxel_downsample.restype = C.c_size_t
        self.lib.orc_voxel_downsample.argtypes = [_V, _V, _V, C.c_size_t, C.c_float, _V, _V, _V]
        nv = self.lib.orc_voxel_downsample(_p(x), _p(y), _p(z), n, voxel_size, *[_p(o) for o in out])
        if nv == C.c_size_t(-1).value:
            raise ValueError("voxel_size must be > 0 and finite")
        return tuple(o[:nv].copy() for o in out)

    def ransac_plane_samples(self, x, y, z, samples, thr):
        x, y, z = map(_f, (x, y, z))
        n = len(x)
        smp = np.ascontiguousarray(np.asarray(samples, dtype=np.uint32).reshape(-1, 3))
        model = (C.c_float * 4)()
        mask = np.zeros(max(n, 1), dtype=np.uint8)
        best = C.c_uint64(0)
        self.lib.orc_ransac_plane_samples.restype = C.c_size_t
        self.lib.orc_ransac_plane_samples.argtypes = [_V, _V, _V, C.c_size_t, _V, C.c_size_t,
                                                      C.c_float, _V, _V, _V]
        kept = self.lib.orc_ransac_plane_samples(_p(x), _p(y), _p(z), n, _p(smp), len(smp), thr,
                                                 model, _p(mask), C.byref(best))
        inl = np.nonzero(mask[:n])[0]
        assert len(inl) == kept
        return np.array(model[:3], dtype=np.float32), float(model[3]), inl, int(best.value)

    def euclidean_cluster(self, x, y, z, thr, min_size, max_size):
        x, y, z = map(_f, (x, y, z))
        n = len(x)
        offsets = np.zeros(n + 1, dtype=np.uint32)
        indices = np.zeros(max(n, 1), dtype=np.uint32)
        self.lib.orc_euclidean_cluster.restype = C.c_size_t
        self.lib.orc_euclidean_cluster.argtypes = [_V, _V, _V, C.c_size_t, C.c_float, C.c_size_t,
                                                   C.c_size_t, _V, _V]
        nc = self.lib.orc_euclidean_cluster(_p(x), _p(y), _p(z), n, thr, min_size, max_size,
                                            _p(offsets), _p(indices))
        return [indices[offsets[c]:offsets[c + 1]].copy() for c in range(nc)]

    def eigvec(self, a00, a01, a02, a11, a12, a22):
        out = (C.c_float * 3)()
        self.lib.orc_smallest_eigenvector_3x3(a00, a01, a02, a11, a12, a22, out)
        return np.array(out, dtype=np.float32)

    def normals(self, x, y, z, k, vp=(0, 0, 0), algo=0, threads=0):
        x, y, z = map(_f, (x, y, z))
        n = len(x)
        if n == 0 or k == 0:
            e = np.empty(0, dtype=np.float32)
            return e, e.copy(), e.copy()
        out = [np.empty(n, dtype=np.float32) for _ in range(3)]
        v = (C.c_float * 3)(*[float(t) for t in vp])
        self.lib.orc_normals(_p(x), _p(y), _p(z), n, k, v, algo, threads, *[_p(o) for o in out])
        return tuple(out)

    def apply_transform(self, x, y, z, rot, t):
        x, y, z = map(_f, (x, y, z))
        out = [np.empty(len(x), dtype=np.float32) for _ in range(3)]
        r = (C.c_float * 9)(*[float(v) for v in np.asarray(rot, dtype=np.float32).reshape(9)])
        tt = (C.c_float * 3)(*[float(v) for v in t])
        self.lib.orc_apply_transform(_p(x), _p(y), _p(z), len(x), r, tt, *[_p(o) for o in out])
        return tuple(out)

    def compose(self, ra, ta, rb, tb):
        mk9 = lambda r: (C.c_float * 9)(*[float(v) for v in np.asarray(r, dtype=np.float32).reshape(9)])
        mk3 = lambda t: (C.c_float * 3)(*[float(v) for v in t])
        ro, to = (C.c_float * 9)(), (C.c_float * 3)()
        self.lib.orc_compose(mk9(ra), mk3(ta), mk9(rb), mk3(tb), ro, to)
        return np.array(ro, dtype=np.float32).reshape(3, 3), np.array(to, dtype=np.float32)

    def find_correspondences(self, sx, sy, sz, tx, ty, tz, max_d=np.inf, algo=0, threads=0):
        sx, sy, sz, tx, ty, tz = map(_f, (sx, sy, sz, tx, ty, tz))
        n = len(sx)
        tgt = np.full(n, NO_INDEX, dtype=np.uint32)
        dist = np.full(n, np.nan, dtype=np.float32)
        nc = self.lib.orc_find_correspondences(_p(sx), _p(sy), _p(sz), n, _p(tx), _p(ty), _p(tz),
                                               len(tx), max_d, algo, threads, _p(tgt), _p(dist))
        return tgt, dist, int(nc)

    def solve_plane(self, ata36, atb6):
        a = (C.c_double * 36)(*[float(v) for v in ata36])
        b = (C.c_double * 6)(*[float(v) for v in atb6])
        rot, t = (C.c_float * 9)(), (C.c_float * 3)()
        self.lib.orc_solve_plane_system(a, b, rot, t)
        return np.array(rot, dtype=np.float32).reshape(3, 3), np.array(t, dtype=np.float32)

    def icp_step(self, cur, tgt, tn=None, max_d=np.inf, algo=0, threads=0):
        cx, cy, cz = map(_f, cur)
        tx, ty, tz = map(_f, tgt)
        tnx, tny, tnz = map(_f, tn) if tn is not None else (None, None, None)
        n = len(cx)
        plane = np.zeros(42, dtype=np.float64)
        point = np.zeros(15, dtype=np.float64)
        rmse, sd2 = C.c_float(), C.c_double()
        tgt_idx = np.full(n, NO_INDEX, dtype=np.uint32)
        dist = np.full(n, np.nan, dtype=np.float32)
        nc = self.lib.orc_icp_step(_p(cx), _p(cy), _p(cz), n, _p(tx), _p(ty), _p(tz), _p(tnx),
                                   _p(tny), _p(tnz), len(tx), max_d, algo, threads, _p(plane),
                                   _p(point), C.byref(rmse), C.byref(sd2), _p(tgt_idx), _p(dist))
        return dict(n_corr=int(nc), plane=plane, point=point, rmse=rmse.value, sum_d2=sd2.value,
                    tgt_idx=tgt_idx, dist=dist)

    def icp(self, mode, src, tgt, tn=None, n_normals=None, max_iter=50, tol=1e-5, max_d=np.inf,
            algo=0, threads=0, with_trace=False):
        sx, sy, sz = map(_f, src)
        tx, ty, tz = map(_f, tgt)
        tnx, tny, tnz = map(_f, tn) if tn is not None else (None, None, None)
        if n_normals is None:
            n_normals = len(tnx) if tnx is not None else 0
        res = IcpResult()
        trace = np.full(2 * max(max_iter, 1), np.nan, dtype=np.float32) if with_trace else None
        rc = self.lib.orc_icp(mode, _p(sx), _p(sy), _p(sz), len(sx), _p(tx), _p(ty), _p(tz),
                              _p(tnx), _p(tny), _p(tnz), len(tx), n_normals, max_iter, tol, max_d,
                              algo, threads, C.byref(res), _p(trace))
        return rc, res, trace


_cached = None


def load():
    global _cached
    if _cached is not None:
        return _cached
    src = os.path.join(ORACLE_DIR, "pcrs_oracle.c")
    if not os.path.exists(LIB) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(LIB)):
        subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    _cached = Oracle(C.CDLL(LIB))
    return _cached
