"""ctypes binding of libpcrs_hip.so (declarations follow include/pcrs_hip.h one to one).

There is no CPU fallback: if the library is missing, or no GPU is present when a
context is created, the call raises.
"""
from __future__ import annotations

import ctypes as C
import sys
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# PCRS_HIP_LIB names another build of the same library (used to compare kernel variants)
LIB_PATH = os.environ.get("PCRS_HIP_LIB") or os.path.normpath(
    os.path.join(_HERE, "..", "lib", "libpcrs_hip.so"))

PCRS_OK = 0
PCRS_ERR_INVALID_ARG = -1
PCRS_ERR_HIP = -2
PCRS_ERR_OOM = -3
PCRS_ERR_NORMALS_MISMATCH = -4
NO_INDEX = 0xFFFFFFFF
ICP_POINT_TO_POINT = 0
ICP_POINT_TO_PLANE = 1
ALGO_AUTO, ALGO_BRUTE, ALGO_GRID = 0, 1, 2
ICP_NSUMS = 32

c_float_p = C.POINTER(C.c_float)
c_u32_p = C.POINTER(C.c_uint32)
c_u8_p = C.POINTER(C.c_uint8)
c_f64_p = C.POINTER(C.c_double)
c_size_p = C.POINTER(C.c_size_t)


class IcpResultC(C.Structure):
    _fields_ = [
        ("rotation", C.c_float * 9),
        ("translation", C.c_float * 3),
        ("fitness", C.c_float),
        ("rmse", C.c_float),
        ("converged", C.c_int32),
        ("num_iterations", C.c_uint32),
    ]


class ObstacleParamsC(C.Structure):
    _fields_ = [("voxel_size", C.c_float), ("sor_k", C.c_uint32), ("sor_std_mul", C.c_float),
                ("plane_threshold", C.c_float), ("plane_iterations", C.c_uint32),
                ("plane_seed", C.c_uint64), ("cluster_threshold", C.c_float),
                ("cluster_min_size", C.c_size_t), ("cluster_max_size", C.c_size_t)]


class ObstacleResultC(C.Structure):
    _fields_ = [("n_voxels", C.c_size_t), ("n_kept", C.c_size_t), ("n_ground", C.c_size_t),
                ("n_obstacles", C.c_size_t), ("n_clusters", C.c_size_t),
                ("n_cluster_points", C.c_size_t), ("plane", C.c_float * 4),
                ("sor_threshold", C.c_float)]


class DistSorResultC(C.Structure):
    _fields_ = [("n_answered", C.c_size_t), ("n_keep", C.c_size_t), ("n_escalated", C.c_size_t),
                ("threshold", C.c_float), ("has_finite", C.c_int), ("by_slabs", C.c_int)]


DIST_UNIQUE_ID_BYTES = 128

# name -> (restype, argtypes); every symbol include/pcrs_hip.h declares
_V = C.c_void_p
_F3 = [_V, _V, _V]  # three column pointers (host or device)
SIGNATURES = {
    "pcrs_ctx_create": (C.c_int, [C.c_int, C.POINTER(_V)]),
    "pcrs_ctx_set_stream": (C.c_int, [_V, _V]),
    "pcrs_ctx_destroy": (None, [_V]),
    "pcrs_last_error": (C.c_char_p, [_V]),
    "pcrs_ctx_synchronize": (C.c_int, [_V]),
    "pcrs_version": (C.c_char_p, []),
    "pcrs_index_build": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_uint32, C.c_int, C.POINTER(_V)]),
    "pcrs_index_build_dev": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_uint32, C.c_int, C.POINTER(_V)]),
    "pcrs_index_destroy": (None, [_V]),
    "pcrs_index_len": (C.c_size_t, [_V]),
    "pcrs_knn_batch": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_uint32, _V, _V, _V]),
    "pcrs_knn_batch_dev": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_uint32, _V, _V, _V]),
    "pcrs_radius_count": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_float, _V]),
    "pcrs_radius_count_dev": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_float, _V]),
    "pcrs_radius_search": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_float, _V, _V, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pcrs_radius_search_dev": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_float, _V, _V, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pcrs_radius_outlier": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_float, C.c_size_t, _V, _V,
                                                   c_size_p]),
    "pcrs_voxel_downsample": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_float] + _F3 + [c_size_p]),
    "pcrs_voxel_downsample_dev": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_float] + _F3 + [c_size_p]),
    "pcrs_ransac_plane_samples": (C.c_int, [_V] + _F3 + [C.c_size_t, _V, C.c_size_t, C.c_float, _V,
                                                         _V, c_size_p, _V]),
    "pcrs_ransac_plane_samples_dev": (C.c_int, [_V] + _F3 + [C.c_size_t, _V, C.c_size_t, C.c_float,
                                                             _V, _V, c_size_p, _V]),
    "pcrs_euclidean_cluster": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_float, C.c_size_t, C.c_size_t,
                                                      _V, _V, c_size_p]),
    "pcrs_euclidean_cluster_dev": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_float, C.c_size_t,
                                                          C.c_size_t, _V, _V, c_size_p, c_size_p]),
    "pcrs_sor_select_dev": (C.c_int, [_V, _V, C.c_size_t, C.c_float, _V, C.c_size_t, _V,
                                      C.POINTER(C.c_float), C.POINTER(C.c_int), c_size_p]),
    "pcrs_sor_select_static_dev": (C.c_int, [_V, _V, C.c_size_t, C.c_float, _V, C.c_size_t, _V, _V]),
    "pcrs_radius_outlier_dev": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_float, C.c_size_t, _V, _V,
                                                       c_size_p]),
    "pcrs_sor_mean_dist_dev": (C.c_int, [_V, C.c_size_t, C.c_size_t, C.c_uint32, _V]),
    "pcrs_sor_threshold_dev": (C.c_int, [_V, _V, C.c_size_t, C.c_float, c_float_p, C.POINTER(C.c_int)]),
    "pcrs_sor_mask_dev": (C.c_int, [_V, _V, C.c_size_t, C.c_float, _V, c_size_p]),
    "pcrs_sor": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_uint32, C.c_float, _V, c_size_p, c_float_p, _V]),
    "pcrs_normals_dev": (C.c_int, [_V, C.c_size_t, C.c_size_t, C.c_uint32, c_float_p, _V, _V, _V]),
    "pcrs_sor_normals_dev": (C.c_int, [_V, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, c_float_p, _V, _V, _V, _V]),
    "pcrs_index_slot_bounds": (C.c_int, [_V, C.c_uint32, c_size_p]),
    "pcrs_sor_normals_slots_dev": (C.c_int, [_V, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32,
                                             c_float_p, _V, _V, _V, _V, _V]),
    "pcrs_scatter_f32_dev": (C.c_int, [_V, _V, _V, C.c_size_t, _V]),
    "pcrs_select_dev": (C.c_int, [_V] + _F3 + [C.c_size_t, _V, C.c_int, _V, _V, _V, _V, c_size_p]),
    "pcrs_sample_triples": (C.c_int, [C.c_size_t, C.c_size_t, C.c_uint64, _V, c_size_p]),
    "pcrs_obstacle_pipeline": (C.c_int, [_V] + _F3 + [C.c_size_t, C.POINTER(ObstacleParamsC),
                                                      C.POINTER(ObstacleResultC), _V, _V, _V, _V, _V]),
    "pcrs_route_pairs_dev": (C.c_int, [_V, _V, _V, C.c_size_t, C.c_size_t, C.c_uint32, _V, _V, c_size_p]),
    "pcrs_scatter_range_f32_dev": (C.c_int, [_V, _V, _V, C.c_size_t, C.c_size_t, _V]),
    "pcrs_slab_cuts_dev": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(C.c_int),
                                                  c_float_p, c_float_p, c_size_p]),
    "pcrs_index_build_box_dev": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_uint32, c_float_p, c_float_p,
                                                        C.POINTER(_V)]),
    "pcrs_index_select_box": (C.c_int, [_V, c_float_p, c_float_p, c_size_p]),
    "pcrs_slab_index_build_dev": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                         C.POINTER(_V), c_size_p, C.POINTER(C.c_int),
                                                         c_float_p]),
    "pcrs_sor_normals_selected_dev": (C.c_int, [_V, C.c_uint32, C.c_uint32, c_float_p, _V, _V, _V, _V,
                                                _V, c_size_p]),
    "pcrs_normals": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_uint32, c_float_p, _V, _V, _V]),
    "pcrs_sor_normals": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_uint32, C.c_float, C.c_uint32, c_float_p, _V, c_size_p, c_float_p, _V, _V, _V]),
    "pcrs_apply_transform": (C.c_int, [_V] + _F3 + [C.c_size_t, c_float_p, c_float_p, _V, _V, _V]),
    "pcrs_apply_transform_dev": (C.c_int, [_V] + _F3 + [C.c_size_t, c_float_p, c_float_p, _V, _V, _V]),
    "pcrs_compose": (None, [c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p]),
    "pcrs_find_correspondences": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_float, _V, _V, c_size_p]),
    "pcrs_find_correspondences_dev": (C.c_int, [_V] + _F3 + [C.c_size_t, C.c_float, _V, _V, c_size_p]),
    "pcrs_icp_create": (C.c_int, [_V] + _F3 + [C.c_size_t] + _F3 + [C.c_size_t] + _F3 + [C.c_size_t, C.POINTER(_V)]),
    "pcrs_icp_create_dev": (C.c_int, [_V] + _F3 + [C.c_size_t] + _F3 + [C.c_size_t] + _F3 + [C.c_size_t, C.POINTER(_V)]),
    "pcrs_icp_destroy": (None, [_V]),
    "pcrs_icp_reset": (C.c_int, [_V]),
    "pcrs_icp_accumulate": (C.c_int, [_V, C.c_int, C.c_float, c_f64_p]),
    "pcrs_icp_solve": (C.c_int, [C.c_int, c_f64_p, c_float_p, c_float_p]),
    "pcrs_icp_apply": (C.c_int, [_V, c_float_p, c_float_p]),
    "pcrs_icp_get_current": (C.c_int, [_V, _V, _V, _V]),
    "pcrs_icp_run": (C.c_int, [_V, C.c_int, C.c_uint32, C.c_float, C.c_float, C.POINTER(IcpResultC)]),
    "pcrs_icp_shard_begin": (C.c_int, [_V, C.c_size_t]),
    "pcrs_icp_shard_accumulate_dev": (C.c_int, [_V, C.c_int, C.c_float, C.c_uint32, _V, C.POINTER(_V)]),
    "pcrs_icp_shard_update_dev": (C.c_int, [_V, C.c_int, _V, C.c_uint32, C.c_float, _V, C.c_int]),
    "pcrs_icp_shard_rmse_dev": (C.c_int, [_V, _V]),
    "pcrs_icp_shard_state": (C.c_int, [_V, C.POINTER(IcpResultC), C.POINTER(C.c_int), C.POINTER(C.c_uint32)]),
    "pcrs_icp_shard_finish": (C.c_int, [_V]),
    "pcrs_icp_register": (C.c_int, [_V, C.c_int] + _F3 + [C.c_size_t] + _F3 + [C.c_size_t] + _F3 + [C.c_size_t, C.c_uint32, C.c_float, C.c_float, C.POINTER(IcpResultC)]),
    "pcrs_dist_unique_id": (C.c_int, [C.c_char_p]),
    "pcrs_dist_comm_init": (C.c_int, [_V, C.c_char_p, C.c_int, C.c_int, C.POINTER(_V)]),
    "pcrs_dist_comm_destroy": (None, [_V]),
    "pcrs_dist_sor_normals_dev": (C.c_int, [_V, _V, C.c_int, C.c_int] + _F3 + [
        C.c_size_t, C.c_size_t, C.c_uint32, C.c_float, C.c_uint32, c_float_p, C.c_size_t,
        _V, _V, _V, _V, _V, _V, C.POINTER(DistSorResultC)]),
    "pcrs_dist_icp_run": (C.c_int, [_V, _V, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_uint32,
                                    C.c_float, C.c_float, C.POINTER(IcpResultC)]),
    "pcrs_index_last_stats": (C.c_int, [_V, C.POINTER(C.c_uint64)]),
    "pcrs_index_levels": (C.c_int, [_V, c_u32_p, c_u32_p]),
    "pcrs_ctx_timing": (C.c_int, [_V, C.c_int]),
    "pcrs_ctx_timing_read": (C.c_int, [_V, c_f64_p, C.POINTER(C.c_uint64)]),
}

_lib = None
_lib_lock = threading.Lock()


def load_library():
    """Loads libpcrs_hip.so (built in-tree by `make -C csrc`) and declares every entry point."""
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C pointclouds-rs_amd/csrc` "
                "(there is no CPU fallback)"
            )
        # PyTorch-ROCm ships its own copy of the HIP runtime.  When both end up in one process the
        # copy loaded second finds no GPU, so torch (when installed) goes first and the library
        # then binds to the runtime that is already there.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        lib = C.CDLL(LIB_PATH)
        variant = bool(os.environ.get("PCRS_HIP_LIB"))
        for name, (res, args) in SIGNATURES.items():
            if variant and not hasattr(lib, name):
                continue  # an older build under comparison: it lacks the newer entry points
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


class PcrsError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"pcrs error {code}: {message}")
        self.code = code
        self.message = message


class Context:
    """One pcrs_ctx (one device, one stream)."""

    def __init__(self, device: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.pcrs_ctx_create(device, C.byref(h))
        if rc != PCRS_OK:
            raise PcrsError(rc, f"pcrs_ctx_create(device={device}) failed (is a GPU visible?)")
        self.handle = h
        self.device = device

    def check(self, rc: int):
        if rc != PCRS_OK:
            msg = self.lib.pcrs_last_error(self.handle)
            raise PcrsError(rc, msg.decode() if msg else "")

    def set_stream(self, stream_ptr: int):
        self.check(self.lib.pcrs_ctx_set_stream(self.handle, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self.check(self.lib.pcrs_ctx_synchronize(self.handle))

    def close(self):
        if self.handle:
            self.lib.pcrs_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        # at interpreter shutdown the HIP runtime may already be gone: the process frees it all
        if sys is None or sys.is_finalizing():  # module globals are cleared late in shutdown
            return
        try:
            self.close()
        except Exception:
            pass


_contexts = {}


def get_context(device: int = 0) -> Context:
    ctx = _contexts.get(device)
    if ctx is None:
        ctx = Context(device)
        _contexts[device] = ctx
    return ctx
