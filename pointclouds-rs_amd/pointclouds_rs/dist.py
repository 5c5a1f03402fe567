"""Query-sharded SOR + normals and ICP across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Every rank owns a
contiguous index range of the cloud, resident in its HBM:

  1. all-gather the three SoA columns, so every rank holds the whole reference cloud
     (100 M points = 1.2 GB, far below 288 GB);
  2. build the search index over the whole cloud and answer the rank's own queries
     (fused SOR mean distance + normals, one neighbour search);
  3. all-gather the per-point mean distances (4 B/pt; with ownership by cells a sum over ranks
     of vectors that are zero outside the rank's own points); every rank derives the same
     global threshold with the reference's summation order and masks its own points.

Which queries a rank answers is a choice (`ownership`): the points it was handed (results in
input order, but scattered all over the replicated index), or a run of whole cells of the index
(a compact region of space; the results carry their global point indices).

ICP shards the source; each iteration all-reduces the PCRS_ICP_NSUMS f64 partial sums and every
rank solves the same small system, so no transform broadcast is needed.

The numeric work is done by an `engine` object; HipEngine drives libpcrs_hip.so through its
device-pointer entry points.  (Tests of the sharding logic inject a CPU engine.)
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _ffi

_V = C.c_void_p


def shard_range(n_total: int, world: int, rank: int):
    """Contiguous [begin, end) of rank's queries; concatenating ranks restores input order."""
    base, rem = divmod(n_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class HipEngine:
    """Device-pointer entry points of libpcrs_hip.so over torch CUDA tensors."""

    def __init__(self, device_index: int = 0, use_torch_stream: bool = True):
        import torch

        self.torch = torch
        self.device = torch.device("cuda", device_index)
        # a context of its own: pointing the process-wide context of pointclouds_rs.hip at
        # torch's stream would silently move every later hip.* call onto that stream too
        self.ctx = _ffi.Context(device_index)
        if use_torch_stream:
            self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.lib = self.ctx.lib

    _shared = {}

    @classmethod
    def shared(cls, device_index: int = 0):
        """One engine per device for callers that create engines over and over (test suites,
        drivers): every engine owns a context with its own pooled device memory, and two engines
        on one GPU would also share the static work space of pcrs_sor_select_static_dev."""
        eng = cls._shared.get(device_index)
        if eng is None or eng.ctx is None:
            eng = cls._shared[device_index] = cls(device_index)
        return eng

    def close(self):
        """Releases the engine's context (stream binding, pinned block, pooled device memory)."""
        ctx, self.ctx = self.ctx, None
        if ctx is not None:
            ctx.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # -- tensors
    def empty(self, n, dtype="float32"):
        return self.torch.empty(n, dtype=getattr(self.torch, dtype), device=self.device)

    def from_numpy(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    @staticmethod
    def _p(t):
        return _V(t.data_ptr()) if t is not None and t.numel() else _V(None)

    # -- index
    def build_index(self, x, y, z, k_hint=0, algo=_ffi.ALGO_AUTO):
        h = _V()
        self.ctx.check(self.lib.pcrs_index_build_dev(self.ctx.handle, self._p(x), self._p(y),
                                                     self._p(z), x.numel(), k_hint, algo,
                                                     C.byref(h)))
        return h

    def destroy_index(self, h):
        self.lib.pcrs_index_destroy(h)

    def index_stats(self, h):
        s = (C.c_uint64 * 3)()
        self.ctx.check(self.lib.pcrs_index_last_stats(h, s))
        return [int(v) for v in s]

    # -- fused consumers over a query range
    def sor_normals_range(self, index, q_begin, q_count, k_sor, k_nrm, viewpoint, mean, nx, ny, nz):
        vp = (C.c_float * 3)(*[float(v) for v in viewpoint])
        self.ctx.check(self.lib.pcrs_sor_normals_dev(index, q_begin, q_count, k_sor, k_nrm, vp,
                                                     self._p(mean), self._p(nx), self._p(ny),
                                                     self._p(nz)))

    # -- the same over runs of cell-ordered slots (ownership by space)
    def slot_bounds(self, index, parts):
        """[bounds[0], ..., bounds[parts]] of the runs of whole cells; all 0 when the index has
        no cell order to split."""
        b = (C.c_size_t * (parts + 1))()
        self.ctx.check(self.lib.pcrs_index_slot_bounds(index, parts, b))
        return [int(v) for v in b]

    def sor_normals_slots(self, index, s_begin, s_count, k_sor, k_nrm, viewpoint, point, mean,
                          nx, ny, nz):
        vp = (C.c_float * 3)(*[float(v) for v in viewpoint])
        self.ctx.check(self.lib.pcrs_sor_normals_slots_dev(index, s_begin, s_count, k_sor, k_nrm,
                                                           vp, self._p(point), self._p(mean),
                                                           self._p(nx), self._p(ny), self._p(nz)))

    # -- ownership by slabs of space: an index over the rank's slab plus a halo
    def slab_cuts(self, x, y, z, parts, k_hint):
        """(axis, cuts[parts + 1], halo, n_finite) of the cloud in the device columns x, y, z."""
        axis = C.c_int()
        cuts = (C.c_float * (parts + 1))()
        halo = C.c_float()
        nf = C.c_size_t()
        self.ctx.check(self.lib.pcrs_slab_cuts_dev(self.ctx.handle, self._p(x), self._p(y), self._p(z),
                                                   x.numel(), parts, k_hint, C.byref(axis), cuts,
                                                   C.byref(halo), C.byref(nf)))
        return int(axis.value), [float(c) for c in cuts], float(halo.value), int(nf.value)

    def build_index_box(self, x, y, z, k_hint, box_lo, box_hi):
        h = _V()
        lo = (C.c_float * 3)(*box_lo)
        hi = (C.c_float * 3)(*box_hi)
        self.ctx.check(self.lib.pcrs_index_build_box_dev(self.ctx.handle, self._p(x), self._p(y),
                                                         self._p(z), x.numel(), k_hint, lo, hi,
                                                         C.byref(h)))
        return h

    def slab_index(self, x, y, z, k_hint, parts, part):
        """Cut, index over slab `part` and its halo, selection of the slab's points, in one call:
        (index, n_queries, axis) or (None, 0, axis) when the cloud has no slabs."""
        h = _V()
        n = C.c_size_t()
        axis = C.c_int()
        slab = (C.c_float * 2)()
        self.ctx.check(self.lib.pcrs_slab_index_build_dev(self.ctx.handle, self._p(x), self._p(y),
                                                          self._p(z), x.numel(), k_hint, parts, part,
                                                          C.byref(h), C.byref(n), C.byref(axis), slab))
        return (h if h.value else None), int(n.value), int(axis.value)

    def select_box(self, index, q_lo, q_hi):
        """Number of indexed points inside [q_lo, q_hi): the queries of sor_normals_selected.
        None when the box holds too few points for a grid."""
        n = C.c_size_t()
        rc = self.lib.pcrs_index_select_box(index, (C.c_float * 3)(*q_lo), (C.c_float * 3)(*q_hi),
                                            C.byref(n))
        if rc == _ffi.PCRS_ERR_INVALID_ARG:
            return None
        self.ctx.check(rc)
        return int(n.value)

    def sor_normals_selected(self, index, k_sor, k_nrm, viewpoint, point, mean, nx, ny, nz):
        """Returns how many of the selected queries the box could not certify (answered from an
        index over the whole cloud)."""
        vp = (C.c_float * 3)(*[float(v) for v in viewpoint])
        esc = C.c_size_t()
        self.ctx.check(self.lib.pcrs_sor_normals_selected_dev(index, k_sor, k_nrm, vp, self._p(point),
                                                              self._p(mean), self._p(nx),
                                                              self._p(ny), self._p(nz), C.byref(esc)))
        return int(esc.value)

    def route_pairs(self, point, values, n_total, world, send_point, send_values):
        """Buckets the (point, value) pairs by the rank that owns the point's index range; returns
        the bucket sizes (the send buffers then hold the buckets in rank order)."""
        counts = (C.c_size_t * world)()
        self.ctx.check(self.lib.pcrs_route_pairs_dev(self.ctx.handle, self._p(point), self._p(values),
                                                     point.numel(), n_total, world,
                                                     self._p(send_point), self._p(send_values), counts))
        return [int(c) for c in counts]

    def scatter_range(self, values, point, first, out):
        """out[point[i] - first] = values[i]"""
        self.ctx.check(self.lib.pcrs_scatter_range_f32_dev(self.ctx.handle, self._p(values),
                                                           self._p(point), values.numel(), first,
                                                           self._p(out)))

    def scatter(self, values, point, out):
        """out[point[i]] = values[i]"""
        self.ctx.check(self.lib.pcrs_scatter_f32_dev(self.ctx.handle, self._p(values),
                                                     self._p(point), values.numel(),
                                                     self._p(out)))

    def threshold(self, mean_full, std_mul):
        thr = C.c_float()
        ok = C.c_int()
        self.ctx.check(self.lib.pcrs_sor_threshold_dev(self.ctx.handle, self._p(mean_full),
                                                       mean_full.numel(), float(std_mul),
                                                       C.byref(thr), C.byref(ok)))
        return float(thr.value), bool(ok.value)

    def select(self, mean_full, std_mul, mean, keep):
        """Threshold over mean_full and mask of `mean` with a single read-back:
        (thr, has_finite, n_keep)."""
        thr = C.c_float()
        ok = C.c_int()
        nk = C.c_size_t(0)
        self.ctx.check(self.lib.pcrs_sor_select_dev(self.ctx.handle, self._p(mean_full),
                                                    mean_full.numel(), float(std_mul),
                                                    self._p(mean), mean.numel(), self._p(keep),
                                                    C.byref(thr), C.byref(ok), C.byref(nk)))
        return float(thr.value), bool(ok.value), int(nk.value)

    def select_static(self, mean_full, std_mul, mean, keep, result):
        """pcrs_sor_select_static_dev: the same selection without any allocation or
        synchronisation; `result` is a device tensor of 16 bytes that receives
        {f32 thr, i32 has_finite, u64 n_keep}."""
        self.ctx.check(self.lib.pcrs_sor_select_static_dev(
            self.ctx.handle, self._p(mean_full), mean_full.numel(), float(std_mul), self._p(mean),
            mean.numel(), self._p(keep), self._p(result)))

    def mask(self, mean, thr, keep, want_count=True):
        nk = C.c_size_t(0)
        self.ctx.check(self.lib.pcrs_sor_mask_dev(self.ctx.handle, self._p(mean), mean.numel(),
                                                  float(thr), self._p(keep),
                                                  C.byref(nk) if want_count else None))
        return int(nk.value)

    # -- timing of the neighbour-search kernel
    def timing(self, enable):
        self.ctx.check(self.lib.pcrs_ctx_timing(self.ctx.handle, 1 if enable else 0))

    def timing_read(self):
        ms = C.c_double()
        n = C.c_uint64()
        self.ctx.check(self.lib.pcrs_ctx_timing_read(self.ctx.handle, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    # -- ICP
    def icp_create(self, src, tgt, normals):
        h = _V()
        nn = normals if normals is not None else (None, None, None)
        n_normals = normals[0].numel() if normals is not None else 0
        self.ctx.check(self.lib.pcrs_icp_create_dev(
            self.ctx.handle, *[self._p(c) for c in src], src[0].numel(),
            *[self._p(c) for c in tgt], tgt[0].numel(), *[self._p(c) for c in nn], n_normals,
            C.byref(h)))
        return h

    def icp_destroy(self, h):
        self.lib.pcrs_icp_destroy(h)

    def icp_reset(self, h):
        self.ctx.check(self.lib.pcrs_icp_reset(h))

    def icp_accumulate(self, h, mode, max_d):
        sums = np.zeros(_ffi.ICP_NSUMS, dtype=np.float64)
        self.ctx.check(self.lib.pcrs_icp_accumulate(h, mode, float(max_d),
                                                    sums.ctypes.data_as(_ffi.c_f64_p)))
        return sums

    def icp_solve(self, mode, sums):
        rot = (C.c_float * 9)()
        t = (C.c_float * 3)()
        rc = self.lib.pcrs_icp_solve(mode, sums.ctypes.data_as(_ffi.c_f64_p), rot, t)
        if rc != 0:
            raise _ffi.PcrsError(rc, "pcrs_icp_solve")
        return rot, t

    def icp_apply(self, h, rot, t):
        self.ctx.check(self.lib.pcrs_icp_apply(h, rot, t))

    def icp_run(self, h, mode, max_iter, tol, max_d):
        res = _ffi.IcpResultC()
        self.ctx.check(self.lib.pcrs_icp_run(h, mode, max_iter, float(tol), float(max_d),
                                             C.byref(res)))
        return res

    # -- ICP with the source split over ranks: both halves of an iteration are only queued
    def icp_shard_begin(self, h, n_source_total):
        self.ctx.check(self.lib.pcrs_icp_shard_begin(h, n_source_total))

    def icp_shard_accumulate(self, h, mode, max_d, iteration, sums, n_local):
        """Queues search + reduction of this shard into the device tensor `sums`; returns the
        shard's dist*dist of this iteration (input order) as a tensor over the handle's memory."""
        p = _V()
        self.ctx.check(self.lib.pcrs_icp_shard_accumulate_dev(h, mode, float(max_d), iteration,
                                                              self._p(sums), C.byref(p)))
        return self._view_f32(p.value, n_local)

    def _view_f32(self, address, n):
        """Tensor over n floats of device memory the library owns (no copy)."""
        if not n:
            return self.empty(0)

        class _Mem:  # what torch.as_tensor reads from an object that exposes device memory
            __cuda_array_interface__ = dict(shape=(n,), typestr="<f4", data=(address, False),
                                            version=2)

        return self.torch.as_tensor(_Mem(), device=self.device)

    def icp_shard_update(self, h, mode, sums_all, world, tolerance, d2_all, metrics_only=False):
        self.ctx.check(self.lib.pcrs_icp_shard_update_dev(
            h, mode, self._p(sums_all), world, float(tolerance),
            self._p(d2_all) if d2_all is not None else _V(None), 1 if metrics_only else 0))

    def icp_shard_rmse(self, h, d2_all):
        self.ctx.check(self.lib.pcrs_icp_shard_rmse_dev(h, self._p(d2_all)))

    def icp_shard_state(self, h):
        """(result, done, last_live) — waits for everything queued so far."""
        res = _ffi.IcpResultC()
        done = C.c_int()
        live = C.c_uint32()
        self.ctx.check(self.lib.pcrs_icp_shard_state(h, C.byref(res), C.byref(done), C.byref(live)))
        return res, bool(done.value), int(live.value)

    def icp_shard_finish(self, h):
        self.ctx.check(self.lib.pcrs_icp_shard_finish(h))

    # -- the library's own multi-GPU driver (collectives on RCCL inside the call)
    def dist_unique_id(self):
        """The 128 bytes rank 0 hands to every rank of a new communicator."""
        buf = C.create_string_buffer(_ffi.DIST_UNIQUE_ID_BYTES)
        if self.lib.pcrs_dist_unique_id(buf) != 0:
            raise _ffi.PcrsError(_ffi.PCRS_ERR_HIP, "RCCL is not available")
        return buf.raw

    def dist_comm_init(self, unique_id: bytes, rank: int, world: int):
        comm = _V()
        self.ctx.check(self.lib.pcrs_dist_comm_init(self.ctx.handle, unique_id, rank, world,
                                                    C.byref(comm)))
        return comm

    def dist_comm_destroy(self, comm):
        self.lib.pcrs_dist_comm_destroy(comm)

    def dist_sor_normals(self, comm, rank, world, shard, n_total, k_sor, std_mul, k_nrm,
                         viewpoint=(0.0, 0.0, 0.0), capacity=None, outputs=None):
        """pcrs_dist_sor_normals_dev: the same result dict as sor_normals_sharded.  `outputs` may
        carry preallocated tensors (point, mean, keep, nx, ny, nz) of one common length."""
        n_local = shard[0].numel()
        outputs = outputs or {}
        cap = n_total if capacity is None else capacity
        if "point" in outputs:
            cap = min(t.numel() for t in outputs.values())

        def buf(name, dtype="float32"):
            t = outputs.get(name)
            return t if t is not None else self.empty(cap, dtype)

        point = buf("point", "int32")
        mean = buf("mean") if k_sor else None
        keep = buf("keep", "uint8") if k_sor else None
        nx, ny, nz = (buf("nx"), buf("ny"), buf("nz")) if k_nrm else (None,) * 3
        res = _ffi.DistSorResultC()
        vp = (C.c_float * 3)(*viewpoint)
        self.ctx.check(self.lib.pcrs_dist_sor_normals_dev(
            self.ctx.handle, comm, rank, world, self._p(shard[0]), self._p(shard[1]),
            self._p(shard[2]), n_local, n_total, k_sor, std_mul, k_nrm, vp, cap, self._p(point),
            self._p(keep), self._p(mean), self._p(nx), self._p(ny), self._p(nz), C.byref(res)))
        n = int(res.n_answered)
        cut = lambda t: None if t is None else t[:n]
        begin, end = shard_range(n_total, world, rank)
        return dict(mean=cut(mean), keep=cut(keep), nx=cut(nx), ny=cut(ny), nz=cut(nz),
                    point=cut(point), thr=float(res.threshold) if res.has_finite else math.nan,
                    n_keep=int(res.n_keep), escalated=int(res.n_escalated), begin=begin, end=end,
                    ownership="slabs" if res.by_slabs else "index", stats=None)

    def dist_icp_run(self, h, comm, rank, world, n_source_total, mode, max_iter, tol, max_d):
        res = _ffi.IcpResultC()
        self.ctx.check(self.lib.pcrs_dist_icp_run(h, comm, rank, world, n_source_total, mode,
                                                  max_iter, tol, max_d, C.byref(res)))
        return res

    def compose(self, ra, ta, rb, tb):
        ro = (C.c_float * 9)()
        to = (C.c_float * 3)()
        self.lib.pcrs_compose(ra, ta, rb, tb, ro, to)
        return ro, to


class Collectives:
    """torch.distributed wrappers; identity when the job has a single rank."""

    def __init__(self, dist=None, stage_through_host=False):
        self.dist = dist if (dist is not None and dist.is_initialized()) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        # a backend without device collectives (gloo): device tensors go through host memory
        self.stage = stage_through_host

    def all_gather_var(self, torch, local, counts):
        """Concatenation over ranks of 1-D tensors whose lengths are `counts` (known everywhere)."""
        if self.world == 1:
            return local
        if self.stage and local.device.type != "cpu":
            return self.all_gather_var(torch, local.cpu(), counts).to(local.device)
        maxc = max(counts)
        local = local.contiguous()
        if min(counts) == maxc:  # equal shards: one collective straight into the result
            out = torch.empty(self.world * maxc, dtype=local.dtype, device=local.device)
            self.dist.all_gather_into_tensor(out, local)
            return out
        # uneven shards: the collective needs equal sizes, so pad to the largest shard
        padded = torch.zeros(maxc, dtype=local.dtype, device=local.device)
        padded[: local.numel()] = local
        out = torch.empty(self.world * maxc, dtype=local.dtype, device=local.device)
        self.dist.all_gather_into_tensor(out, padded)
        return torch.cat([out[r * maxc: r * maxc + counts[r]] for r in range(self.world)])

    def all_reduce_sum(self, t):
        """In-place sum of a tensor over ranks."""
        if self.world > 1:
            if self.stage and t.device.type != "cpu":
                h = t.cpu()
                self.dist.all_reduce(h)
                t.copy_(h)
            else:
                self.dist.all_reduce(t)

    def all_to_all_var(self, torch, send, send_counts, recv_counts=None):
        """send holds world buckets in rank order (sizes send_counts); bucket r goes to rank r.
        Returns (what the ranks sent here, concatenated in rank order; its per-rank sizes)."""
        if self.world == 1:
            return send[: send_counts[0]], list(send_counts)
        if self.stage and send.device.type != "cpu":
            out, rc = self.all_to_all_var(torch, send.cpu(), send_counts, recv_counts)
            return out.to(send.device), rc
        if recv_counts is None:  # every rank learns what the others send it
            m = torch.tensor(send_counts, dtype=torch.int64, device=send.device)
            allm = torch.empty(self.world * self.world, dtype=torch.int64, device=send.device)
            self.dist.all_gather_into_tensor(allm, m)
            recv_counts = [int(v) for v in allm.view(self.world, self.world)[:, self.rank].tolist()]
        total = sum(send_counts)
        send = send[:total].contiguous()
        if self.dist.get_backend() == "nccl":
            out = torch.empty(sum(recv_counts), dtype=send.dtype, device=send.device)
            self.dist.all_to_all_single(out, send, list(recv_counts), list(send_counts))
            return out, list(recv_counts)
        # a backend without all-to-all (gloo): everybody's buckets are gathered and each rank
        # picks its own out of them
        totals = torch.tensor([total], dtype=torch.int64)
        all_totals = [torch.empty_like(totals) for _ in range(self.world)]
        self.dist.all_gather(all_totals, totals)
        all_totals = [int(t.item()) for t in all_totals]
        everything = self.all_gather_var(torch, send, all_totals)
        m = torch.tensor(send_counts, dtype=torch.int64)
        allm = [torch.empty_like(m) for _ in range(self.world)]
        self.dist.all_gather(allm, m)
        parts, base = [], 0
        for r in range(self.world):
            off = base + int(allm[r][: self.rank].sum().item())
            parts.append(everything[off: off + int(allm[r][self.rank].item())])
            base += all_totals[r]
        return torch.cat(parts), list(recv_counts)

    def all_reduce_min(self, t):
        """In-place minimum of a small tensor over ranks."""
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)

    def barrier(self):
        if self.dist:
            self.dist.barrier()


def sor_normals_sharded(engine, coll, shard, n_total, k_sor, std_mul, k_nrm,
                        viewpoint=(0.0, 0.0, 0.0), outputs=None, algo=_ffi.ALGO_AUTO,
                        want_stats=False, ownership="index"):
    """SOR + normals of a cloud whose points are sharded over ranks in index order.

    shard: (x, y, z) tensors of this rank's points.  Returns a dict with the keep mask, means and
    normals of the points this rank answered, the global threshold and the kept count among them.
    `outputs` may carry preallocated tensors (mean, keep, nx, ny, nz, point) to avoid allocation;
    `want_stats` adds the search-stage counters of this call (a device read-back).

    ownership="index": the rank answers the points it holds, results in their order
    (result["begin"], result["end"] = its index range).
    ownership="cells": the rank answers a run of whole grid cells of the replicated index —
    a compact region of space, so its search touches 1/world of the index instead of all of it.
    result["point"] holds the global index of every answered point (int32 bit pattern of a
    uint32), the other arrays are in the same order; the runs of all ranks partition the cloud.
    Falls back to "index" when the index has no cell order (tiny or non-finite clouds);
    result["ownership"] tells which one ran.
    """
    torch = engine.torch
    world, rank = coll.world, coll.rank
    counts = [shard_range(n_total, world, r)[1] - shard_range(n_total, world, r)[0]
              for r in range(world)]
    begin, end = shard_range(n_total, world, rank)
    assert shard[0].numel() == end - begin, "shard length does not match its index range"
    assert ownership in ("index", "cells", "slabs")
    full = [coll.all_gather_var(torch, c, counts) for c in shard]
    if ownership == "slabs":
        res = _sor_normals_slabs(engine, coll, full, n_total, k_sor, std_mul, k_nrm, viewpoint,
                                 outputs if outputs is not None else {}, want_stats)
        if res is not None:
            res.update(begin=begin, end=end)
            return res
        ownership = "index"  # no slabs to cut (non-finite points, a box too small for a grid)
    if outputs is None:
        outputs = {}

    def buf(name, n, dtype="float32"):
        t = outputs.get(name)
        return t[:n] if t is not None and t.numel() >= n else engine.empty(n, dtype)

    want_n = k_nrm > 0
    k_hint = max(k_sor + 1 if k_sor else 0, k_nrm)
    index = engine.build_index(full[0], full[1], full[2], k_hint, algo)
    try:
        bounds = engine.slot_bounds(index, world) if ownership == "cells" else None
        if bounds is not None and bounds[-1] != n_total:
            bounds, ownership = None, "index"
        if bounds is None:
            nq = end - begin
            own_counts = counts
        else:
            own_counts = [bounds[r + 1] - bounds[r] for r in range(world)]
            nq = own_counts[rank]
        # preallocated outputs are sized for the largest run a rank can get
        mean, keep = buf("mean", nq), buf("keep", nq, "uint8")
        nx = ny = nz = point = None
        if want_n:
            nx, ny, nz = buf("nx", nq), buf("ny", nq), buf("nz", nq)
        if bounds is None:
            engine.sor_normals_range(index, begin, nq, k_sor, k_nrm, viewpoint,
                                     mean if k_sor else None, nx, ny, nz)
        else:
            point = buf("point", nq, "int32")
            engine.sor_normals_slots(index, bounds[rank], nq, k_sor, k_nrm, viewpoint, point,
                                     mean if k_sor else None, nx, ny, nz)
        # reading the stage counters back waits for the search: only on request
        stats = engine.index_stats(index) if want_stats else None
    finally:
        engine.destroy_index(index)
    result = dict(mean=mean, keep=keep, nx=nx, ny=ny, nz=nz, thr=math.nan, n_keep=0, stats=stats,
                  begin=begin, end=end, point=point, ownership=ownership)
    if k_sor:
        # the threshold is a sequential sum over the means in index order
        if bounds is None:
            mean_full = coll.all_gather_var(torch, mean, own_counts)
        else:
            # every rank drops its means at their indices into a zeroed vector; the sum over
            # ranks (x + 0 + ... + 0, exact in any order) is the vector of all means
            mean_full = buf("mean_by_index", n_total)
            mean_full.zero_()
            engine.scatter(mean, point, mean_full)
            coll.all_reduce_sum(mean_full)
        thr, ok, n_keep = engine.select(mean_full, std_mul, mean, keep)
        if ok:
            result["thr"] = thr
            result["n_keep"] = n_keep
    return result


def _sor_normals_slabs(engine, coll, full, n_total, k_sor, std_mul, k_nrm, viewpoint, outputs,
                       want_stats):
    """ownership="slabs": the cloud is cut along its longest axis into one slab per rank (equal
    point counts; a pure function of the replicated columns, so every rank derives the same cuts
    without a collective).  A rank indexes only its slab plus a halo — not the whole cloud — and
    answers the slab's points; the few queries whose neighbourhood reaches beyond the halo are
    answered from an index over the whole cloud that is built only when there are any.  The means
    travel as (point index, mean) pairs: all-gather, then every rank drops them into index order.
    Returns None when the cloud cannot be cut this way."""
    torch = engine.torch
    world, rank = coll.world, coll.rank
    k_hint = max(k_sor + 1 if k_sor else 0, k_nrm)
    index, nq, axis = engine.slab_index(full[0], full[1], full[2], k_hint, world, rank)
    try:
        # every rank must take the same road: one without a grid sends all back to index ranges
        ok = torch.tensor([0 if index is None else 1], dtype=torch.int32,
                          device="cpu" if coll.stage else engine.device)
        coll.all_reduce_min(ok)
        if int(ok.item()) == 0:
            return None

        def buf(name, n, dtype="float32"):
            t = outputs.get(name)
            return t[:n] if t is not None and t.numel() >= n else engine.empty(n, dtype)

        mean, keep, point = buf("mean", nq), buf("keep", nq, "uint8"), buf("point", nq, "int32")
        nx = ny = nz = None
        if k_nrm > 0:
            nx, ny, nz = buf("nx", nq), buf("ny", nq), buf("nz", nq)
        escalated = engine.sor_normals_selected(index, k_sor, k_nrm, viewpoint, point,
                                                mean if k_sor else None, nx, ny, nz)
        stats = engine.index_stats(index) if want_stats else None
    finally:
        if index is not None:
            engine.destroy_index(index)
    result = dict(mean=mean, keep=keep, nx=nx, ny=ny, nz=nz, thr=math.nan, n_keep=0, stats=stats,
                  point=point, ownership="slabs", escalated=escalated, slab_axis=axis)
    if k_sor:
        # The threshold is a sequential sum over the means in index order.  Every mean first
        # travels to the rank that owns its index range (all-to-all of (index, mean) pairs: 8 bytes
        # a point, each pair over one link), is dropped into place there, and the ranges -- in rank
        # order the whole vector in index order -- are all-gathered (4 bytes a point).
        send_point, send_mean = buf("send_point", nq, "int32"), buf("send_mean", nq)
        send_counts = engine.route_pairs(point, mean, n_total, world, send_point, send_mean)
        recv_point, recv_counts = coll.all_to_all_var(torch, send_point, send_counts)
        recv_mean, _ = coll.all_to_all_var(torch, send_mean, send_counts, recv_counts)
        begin, end = shard_range(n_total, world, rank)
        assert recv_point.numel() == end - begin, "the slabs do not partition the cloud"
        mean_range = buf("mean_range", end - begin)
        engine.scatter_range(recv_mean, recv_point, begin, mean_range)
        range_counts = [shard_range(n_total, world, r)[1] - shard_range(n_total, world, r)[0]
                        for r in range(world)]
        mean_full = coll.all_gather_var(torch, mean_range, range_counts)
        thr, ok, n_keep = engine.select(mean_full, std_mul, mean, keep)
        if ok:
            result["thr"] = thr
            result["n_keep"] = n_keep
    return result


def icp_sharded(engine, coll, mode, src_shard, n_src_total, tgt, tgt_normals, max_iterations=50,
                tolerance=1e-5, max_correspondence_distance=math.inf, batch=4):
    """ICP with the source sharded over ranks (contiguous index ranges, rank order = index order)
    and the target replicated (loop of crates/registration/src/icp.rs:125-206 /
    icp_plane.rs:20-97).

    The loop state lives on every rank's device.  Per iteration a rank queues the search and
    reduction of its shard, the ranks all-gather their PCRS_ICP_NSUMS partial sums (device
    collective), and every rank queues the same tail (add in rank order, metrics, convergence
    test, solve, compose): identical inputs, identical kernels, so all ranks hold the same state
    and take the same decisions.  The host looks at the state once per `batch` iterations;
    iterations queued behind the end of the loop are no-ops.  With a positive tolerance the rmse
    that steers the loop is the reference's own (icp.rs:277-281: dist*dist of all points in index
    order, summed left to right in f32), so the shards' dist*dist are gathered too (4 B/point per
    iteration); otherwise only the reported rmse is, once, after the loop.
    """
    torch = engine.torch
    world = coll.world
    ident_r = (C.c_float * 9)(1, 0, 0, 0, 1, 0, 0, 0, 1)
    ident_t = (C.c_float * 3)(0, 0, 0)
    out = dict(rotation=ident_r, translation=ident_t, fitness=0.0, rmse=0.0, converged=False,
               num_iterations=0)
    if n_src_total == 0 or tgt[0].numel() == 0:
        out["converged"] = n_src_total == 0 and tgt[0].numel() == 0
        return out
    counts = [shard_range(n_src_total, world, r)[1] - shard_range(n_src_total, world, r)[0]
              for r in range(world)]
    n_local = src_shard[0].numel()
    assert n_local == counts[coll.rank], "shard length does not match its index range"
    exact = tolerance > 0
    h = engine.icp_create(src_shard, tgt, tgt_normals)
    try:
        engine.icp_shard_begin(h, n_src_total)
        sums = engine.empty(_ffi.ICP_NSUMS, "float64")
        d2_of = {}  # iteration -> this shard's dist*dist (two alternating buffers of the handle)

        def iteration(it, metrics_only=False):
            d2_of[it & 1] = engine.icp_shard_accumulate(h, mode, max_correspondence_distance, it,
                                                        sums, n_local)
            sums_all = coll.all_gather_var(torch, sums, [_ffi.ICP_NSUMS] * world)
            d2_all = coll.all_gather_var(torch, d2_of[it & 1], counts) if exact or metrics_only \
                else None
            engine.icp_shard_update(h, mode, sums_all, world, tolerance, d2_all, metrics_only)

        if max_iterations == 0:  # icp.rs:190-197: one metrics pass, identity transform
            iteration(0, metrics_only=True)
        it = 0
        res, done, last_live = None, False, 0
        while it < max_iterations and not done:
            nb = min(batch, max_iterations - it)
            for j in range(nb):
                iteration(it + j)
            it += nb
            res, done, last_live = engine.icp_shard_state(h)
        engine.icp_shard_finish(h)
        if max_iterations > 0 and not exact and last_live:
            # the reported rmse from the reference's sum over the last live iteration
            d2_all = coll.all_gather_var(torch, d2_of[(last_live - 1) & 1], counts)
            engine.icp_shard_rmse(h, d2_all)
        res, done, last_live = engine.icp_shard_state(h)
        out.update(rotation=res.rotation, translation=res.translation, fitness=float(res.fitness),
                   rmse=float(res.rmse), converged=bool(res.converged),
                   num_iterations=int(res.num_iterations))
        return out
    finally:
        engine.icp_destroy(h)
