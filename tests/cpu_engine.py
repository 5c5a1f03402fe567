"""CPU stand-in for pointclouds_rs.dist.HipEngine, backed by the oracle.  TEST INFRASTRUCTURE:
it lets the rank/shard/collective logic of pointclouds_rs.dist run under gloo without a GPU."""
import ctypes as C

import numpy as np
import torch

from pointclouds_rs import _ffi


class CpuEngine:
    def __init__(self, oracle):
        self.torch = torch
        self.device = torch.device("cpu")
        self.o = oracle
        self.lib = _ffi.load_library()  # pcrs_icp_solve / pcrs_compose are pure host math

    def empty(self, n, dtype="float32"):
        return torch.empty(n, dtype=getattr(torch, dtype))

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def build_index(self, x, y, z, k_hint=0, algo=0):
        return (x.numpy(), y.numpy(), z.numpy())

    def destroy_index(self, h):
        pass

    def index_stats(self, h):
        return [0, 0, 0]

    def sor_normals_range(self, index, q_begin, q_count, k_sor, k_nrm, viewpoint, mean, nx, ny, nz):
        x, y, z = index
        sl = slice(q_begin, q_begin + q_count)
        if mean is not None:
            mean.copy_(torch.from_numpy(self.o.sor_mean_dist(x, y, z, k_sor, algo=1)[sl].copy()))
        if nx is not None:
            n = self.o.normals(x, y, z, k_nrm, viewpoint, algo=1)
            for dst, src in zip((nx, ny, nz), n):
                dst.copy_(torch.from_numpy(src[sl].copy()))

    # "cell order" of the stand-in: points sorted by x (cells = unit-wide slabs of x)
    def _order(self, index):
        x = index[0]
        if not np.isfinite(np.stack(index)).all() or len(x) < 64:
            return None
        return np.argsort(x, kind="stable")

    def slot_bounds(self, index, parts):
        order = self._order(index)
        if order is None:
            return [0] * (parts + 1)
        cells = np.floor(index[0][order]).astype(np.int64)
        n = len(order)
        starts = np.flatnonzero(np.r_[True, cells[1:] != cells[:-1]])  # first slot of every cell
        out = []
        for r in range(parts):
            t = r * n // parts
            i = np.searchsorted(starts, t, side="left")
            out.append(int(starts[i]) if i < len(starts) else n)
        return out + [n]

    def sor_normals_slots(self, index, s_begin, s_count, k_sor, k_nrm, viewpoint, point, mean,
                          nx, ny, nz):
        x, y, z = index
        sel = self._order(index)[s_begin:s_begin + s_count]
        point.copy_(torch.from_numpy(sel.astype(np.uint32).view(np.int32).copy()))
        if mean is not None:
            mean.copy_(torch.from_numpy(self.o.sor_mean_dist(x, y, z, k_sor, algo=1)[sel].copy()))
        if nx is not None:
            n = self.o.normals(x, y, z, k_nrm, viewpoint, algo=1)
            for dst, src in zip((nx, ny, nz), n):
                dst.copy_(torch.from_numpy(src[sel].copy()))

    # ---- ownership by slabs of space
    def slab_cuts(self, x, y, z, parts, k_hint):
        X = np.stack([x.numpy(), y.numpy(), z.numpy()], 1)
        fin = np.isfinite(X).all(axis=1)
        n_finite = int(fin.sum())
        cuts = [-np.inf] + [np.inf] * parts
        if n_finite == 0:
            return 0, cuts, 0.0, 0
        P = X[fin]
        lo, hi = P.min(axis=0).astype(np.float64), P.max(axis=0).astype(np.float64)
        ext = hi - lo
        axis = int(np.argmax(ext))
        if not ext[axis] > 0:
            return axis, cuts, 0.0, n_finite
        vol = float(np.prod(np.maximum(ext, 1e-3 * ext[axis])))
        halo = 3.0 * (vol * float(k_hint or 16) / n_finite) ** (1.0 / 3.0)
        v = np.sort(P[:, axis])
        for r in range(1, parts):  # equal counts; points at the same coordinate stay together
            cuts[r] = float(v[r * n_finite // parts])
        return axis, cuts, float(halo), n_finite

    def slab_index(self, x, y, z, k_hint, parts, part):
        axis, cuts, halo, n_finite = self.slab_cuts(x, y, z, parts, k_hint)
        if n_finite != x.numel() or not np.isfinite(cuts[1:parts]).all():
            return None, 0, axis
        inf = float("inf")
        lo, hi, qlo, qhi = [-inf] * 3, [inf] * 3, [-inf] * 3, [inf] * 3
        qlo[axis], qhi[axis] = cuts[part], cuts[part + 1]
        lo[axis] = cuts[part] - halo if part > 0 else -inf
        hi[axis] = cuts[part + 1] + halo if part + 1 < parts else inf
        index = self.build_index_box(x, y, z, k_hint, lo, hi)
        n = self.select_box(index, qlo, qhi)
        return (None, 0, axis) if n is None else (index, n, axis)

    def build_index_box(self, x, y, z, k_hint, box_lo, box_hi):
        return dict(cols=(x.numpy(), y.numpy(), z.numpy()), lo=np.array(box_lo), hi=np.array(box_hi),
                    sel=None)

    def select_box(self, index, q_lo, q_hi):
        X = np.stack(index["cols"], 1)
        inside = np.isfinite(X).all(axis=1)
        boxed = inside.copy()
        for a in range(3):
            inside &= (X[:, a] >= np.float32(q_lo[a])) & (X[:, a] < np.float32(q_hi[a]))
            boxed &= (X[:, a] >= np.float32(index["lo"][a])) & (X[:, a] < np.float32(index["hi"][a]))
        if boxed.sum() < 64:
            return None  # too few points for a grid
        sel = np.flatnonzero(inside)
        index["sel"] = sel[np.argsort(X[sel, 0], kind="stable")]  # some spatial order
        return len(sel)

    def sor_normals_selected(self, index, k_sor, k_nrm, viewpoint, point, mean, nx, ny, nz):
        x, y, z = index["cols"]
        sel = index["sel"]
        point.copy_(torch.from_numpy(sel.astype(np.uint32).view(np.int32).copy()))
        if mean is not None:
            mean.copy_(torch.from_numpy(self.o.sor_mean_dist(x, y, z, k_sor, algo=1)[sel].copy()))
        if nx is not None:
            n = self.o.normals(x, y, z, k_nrm, viewpoint, algo=1)
            for dst, src in zip((nx, ny, nz), n):
                dst.copy_(torch.from_numpy(src[sel].copy()))
        return 0

    def route_pairs(self, point, values, n_total, world, send_point, send_values):
        from pointclouds_rs.dist import shard_range

        p = point.numpy().view(np.uint32)
        ends = np.array([shard_range(n_total, world, r)[1] for r in range(world)])
        dest = np.searchsorted(ends, p, side="right")
        order = np.argsort(dest, kind="stable")
        send_point[: len(p)].copy_(torch.from_numpy(p[order].view(np.int32).copy()))
        send_values[: len(p)].copy_(torch.from_numpy(values.numpy()[order].copy()))
        return [int((dest == r).sum()) for r in range(world)]

    def scatter_range(self, values, point, first, out):
        out.numpy()[point.numpy().view(np.uint32).astype(np.int64) - first] = values.numpy()

    def scatter(self, values, point, out):
        out.numpy()[point.numpy().view(np.uint32)] = values.numpy()

    def threshold(self, mean_full, std_mul):
        ok, thr, _, _ = self.o.sor_threshold(mean_full.numpy(), std_mul)
        return (thr if ok else float("nan")), ok

    def select(self, mean_full, std_mul, mean, keep):
        thr, ok = self.threshold(mean_full, std_mul)
        if not ok:
            keep.zero_()
            return thr, ok, 0
        return thr, ok, self.mask(mean, thr, keep)

    def mask(self, mean, thr, keep, want_count=True):
        k = mean.numpy() <= np.float32(thr)
        keep.copy_(torch.from_numpy(k.astype(np.uint8)))
        return int(k.sum())

    # ---- ICP
    def icp_create(self, src, tgt, normals):
        return dict(src=[c.numpy().copy() for c in src], tgt=[c.numpy() for c in tgt],
                    tn=[c.numpy() for c in normals] if normals is not None else None, cur=None)

    def icp_destroy(self, h):
        pass

    def icp_reset(self, h):
        h["cur"] = list(self.o.apply_transform(*h["src"], np.eye(3), (0, 0, 0)))

    def icp_accumulate(self, h, mode, max_d):
        st = self.o.icp_step(h["cur"], h["tgt"], h["tn"], max_d, algo=1)
        s = np.zeros(_ffi.ICP_NSUMS)
        s[0], s[1] = st["n_corr"], st["sum_d2"]
        if mode == _ffi.ICP_POINT_TO_PLANE:
            ata = st["plane"][:36].reshape(6, 6)
            s[2:23] = [ata[r, c] for r in range(6) for c in range(r, 6)]
            s[23:29] = st["plane"][36:]
        else:
            s[2:17] = st["point"]
        return s

    def icp_solve(self, mode, sums):
        rot, t = (C.c_float * 9)(), (C.c_float * 3)()
        assert self.lib.pcrs_icp_solve(mode, sums.ctypes.data_as(_ffi.c_f64_p), rot, t) == 0
        return rot, t

    def icp_apply(self, h, rot, t):
        h["cur"] = list(self.o.apply_transform(*h["cur"], np.array(rot, np.float32).reshape(3, 3), list(t)))

    # ---- ICP with the source split over ranks: the device-resident loop state of the HIP engine
    # as a dict, the two halves of an iteration as the library documents them
    def _sums(self, h, mode, max_d):
        st = self.o.icp_step(h["cur"], h["tgt"], h["tn"], max_d, algo=1)
        s = np.zeros(_ffi.ICP_NSUMS)
        s[0], s[1] = st["n_corr"], st["sum_d2"]
        if mode == _ffi.ICP_POINT_TO_PLANE:
            ata = st["plane"][:36].reshape(6, 6)
            s[2:23] = [ata[r, c] for r in range(6) for c in range(r, 6)]
            s[23:29] = st["plane"][36:]
        else:
            s[2:17] = st["point"]
        valid = st["tgt_idx"] != 0xFFFFFFFF
        d2 = np.where(valid, st["dist"] * st["dist"], np.float32(0)).astype(np.float32)
        return s, d2

    def icp_shard_begin(self, h, n_source_total):
        self.icp_reset(h)
        eye = (C.c_float * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1)
        h["st"] = dict(total=n_source_total, done=False, converged=False, it=0, cum_r=eye,
                       cum_t=(C.c_float * 3)(), inc=None, prev=np.float32(np.inf),
                       rmse=np.float32(np.inf), fit=np.float32(0), ncorr=np.float32(0), live=0)
        h["d2"] = {}

    def icp_shard_accumulate(self, h, mode, max_d, iteration, sums, n_local):
        st = h["st"]
        if not st["done"]:
            if iteration > 0 and st["inc"] is not None:
                self.icp_apply(h, *st["inc"])
                st["inc"] = None
            s, d2 = self._sums(h, mode, max_d)
            sums.copy_(torch.from_numpy(s))
            h["d2"][iteration & 1] = torch.from_numpy(d2)
        return h["d2"][iteration & 1]

    def icp_shard_update(self, h, mode, sums_all, world, tolerance, d2_all, metrics_only=False):
        st = h["st"]
        if st["done"]:
            return
        s = np.zeros(_ffi.ICP_NSUMS)
        for r in range(world):  # rank order
            s = s + sums_all.numpy()[r * _ffi.ICP_NSUMS:(r + 1) * _ffi.ICP_NSUMS]
        if not metrics_only:
            st["it"] += 1
        if not s[0] > 0:
            st["done"] = True
            return
        ncorr = np.float32(int(s[0]))
        if d2_all is not None:
            total = np.cumsum(d2_all.numpy(), dtype=np.float32)[-1]  # left to right in f32
            rmse = np.sqrt(np.float32(total / ncorr))
        else:
            rmse = np.float32(np.sqrt(s[1] / s[0]))
        st.update(rmse=rmse, fit=np.float32(ncorr / np.float32(st["total"])), ncorr=ncorr,
                  live=1 if metrics_only else st["it"])
        if metrics_only:
            st["done"] = True
            return
        if abs(np.float32(st["prev"] - rmse)) < np.float32(tolerance):
            st["converged"] = st["done"] = True
            return
        st["prev"] = rmse
        st["inc"] = self.icp_solve(mode, s)
        st["cum_r"], st["cum_t"] = self.compose(st["cum_r"], st["cum_t"], *st["inc"])

    def icp_shard_rmse(self, h, d2_all):
        st = h["st"]
        if st["live"]:
            total = np.cumsum(d2_all.numpy(), dtype=np.float32)[-1]
            st["rmse"] = np.sqrt(np.float32(total / st["ncorr"]))

    def icp_shard_state(self, h):
        st = h["st"]
        res = _ffi.IcpResultC()
        res.rotation[:] = list(st["cum_r"])
        res.translation[:] = list(st["cum_t"])
        res.fitness, res.rmse = float(st["fit"]), float(st["rmse"])
        res.converged, res.num_iterations = int(st["converged"]), st["it"]
        return res, st["done"], st["live"]

    def icp_shard_finish(self, h):
        st = h["st"]
        if not st["done"] and st["inc"] is not None:
            self.icp_apply(h, *st["inc"])
            st["inc"] = None

    def compose(self, ra, ta, rb, tb):
        ro, to = (C.c_float * 9)(), (C.c_float * 3)()
        self.lib.pcrs_compose(ra, ta, rb, tb, ro, to)
        return ro, to
