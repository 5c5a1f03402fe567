"""The C boundary must never abort or unwind: wrong arguments come back as status codes with a
message (include/pcrs_hip.h, "Conventions"), results are identical from run to run, and the CPU
oracle itself is clean under AddressSanitizer / UndefinedBehaviorSanitizer."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = np.float32


def test_null_handles_are_rejected_without_a_gpu():
    """Every entry point checks its handle before anything else, so a null context, index or ICP
    object is PCRS_ERR_INVALID_ARG on any machine (no GPU needed to get that far)."""
    from pointclouds_rs import _ffi

    lib = _ffi.load_library()
    assert lib.pcrs_last_error(None) == b"null context"
    buf = (C.c_float * 8)()
    u8 = (C.c_uint8 * 8)()
    u32 = (C.c_uint32 * 16)()
    n_out = C.c_size_t(0)
    thr = C.c_float(0)
    ok = C.c_int(0)
    h = C.c_void_p()
    res = _ffi.IcpResultC()
    f3 = (buf, buf, buf)
    calls = {
        "pcrs_ctx_synchronize": (None,),
        "pcrs_ctx_set_stream": (None, None),
        "pcrs_ctx_timing": (None, 1),
        "pcrs_index_build": (None, *f3, 8, 4, 0, C.byref(h)),
        "pcrs_index_build_dev": (None, *f3, 8, 4, 0, C.byref(h)),
        "pcrs_knn_batch": (None, *f3, 8, 2, u32, buf, u32),
        "pcrs_knn_batch_dev": (None, *f3, 8, 2, u32, buf, u32),
        "pcrs_radius_count": (None, *f3, 8, 1.0, u32),
        "pcrs_radius_outlier": (None, *f3, 8, 1.0, 2, u8, u32, C.byref(n_out)),
        "pcrs_voxel_downsample": (None, *f3, 8, 1.0, *f3, C.byref(n_out)),
        "pcrs_euclidean_cluster": (None, *f3, 8, 1.0, 1, 8, u32, u32, C.byref(n_out)),
        "pcrs_ransac_plane_samples": (None, *f3, 8, u32, 2, 0.1, buf, u8, C.byref(n_out), None),
        "pcrs_sor": (None, *f3, 8, 2, 1.0, u8, C.byref(n_out), C.byref(thr), None),
        "pcrs_normals": (None, *f3, 8, 2, buf, *f3),
        "pcrs_sor_normals": (None, *f3, 8, 2, 1.0, 2, buf, u8, C.byref(n_out), C.byref(thr), *f3),
        "pcrs_sor_threshold_dev": (None, buf, 8, 1.0, C.byref(thr), C.byref(ok)),
        "pcrs_sor_select_dev": (None, buf, 8, 1.0, buf, 8, u8, C.byref(thr), C.byref(ok), C.byref(n_out)),
        "pcrs_sor_select_static_dev": (None, buf, 8, 1.0, buf, 8, u8, u32),
        "pcrs_sor_mean_dist_dev": (None, 0, 8, 2, buf),
        "pcrs_apply_transform": (None, *f3, 8, buf, buf, *f3),
        "pcrs_find_correspondences": (None, *f3, 8, 1.0, u32, buf, C.byref(n_out)),
        "pcrs_icp_create": (None, *f3, 8, *f3, 8, None, None, None, 0, C.byref(h)),
        "pcrs_icp_run": (None, 0, 5, 0.0, 1.0, C.byref(res)),
        "pcrs_icp_reset": (None,),
        "pcrs_icp_register": (None, 0, *f3, 8, *f3, 8, None, None, None, 0, 5, 0.0, 1.0, C.byref(res)),
        "pcrs_radius_search": (None, *f3, 8, 1.0, u32, u32, 8, C.byref(n_out)),
        "pcrs_radius_search_dev": (None, *f3, 8, 1.0, u32, u32, 8, C.byref(n_out)),
        "pcrs_icp_shard_begin": (None, 8),
        "pcrs_icp_shard_accumulate_dev": (None, 1, 1.0, 0, buf, C.byref(h)),
        "pcrs_icp_shard_update_dev": (None, 1, buf, 2, 0.0, None, 0),
        "pcrs_icp_shard_rmse_dev": (None, buf),
        "pcrs_icp_shard_state": (None, C.byref(res), C.byref(ok), None),
        "pcrs_icp_shard_finish": (None,),
    }
    for name, args in calls.items():
        rc = getattr(lib, name)(*args)
        assert rc == _ffi.PCRS_ERR_INVALID_ARG, (name, rc)
    # handles that are allowed to be null where the reference has no failure either
    lib.pcrs_ctx_destroy(None)
    lib.pcrs_index_destroy(None)
    lib.pcrs_icp_destroy(None)
    assert lib.pcrs_index_len(None) == 0


def test_oracle_is_clean_under_sanitizers():
    """`make -C oracle san`: every exported oracle function over empty, tiny, co-planar, lattice
    and non-finite clouds with -fsanitize=address,undefined (host code only)."""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "san"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "calls clean" in r.stdout


@pytest.mark.gpu
def test_absurd_sizes_return_an_error_with_a_message():
    from pointclouds_rs import _ffi

    ctx = _ffi.get_context(0)
    lib = ctx.lib
    x = np.zeros(16, dtype=F)
    p = x.ctypes.data_as(C.c_void_p)
    keep = np.zeros(16, dtype=np.uint8)
    n_out = C.c_size_t(0)
    thr = C.c_float(0)
    for n in (1 << 33, (1 << 64) - 1, 0xFFFFFFFF):
        rc = lib.pcrs_sor(ctx.handle, p, p, p, n, 16, 1.0, keep.ctypes.data_as(C.c_void_p),
                          C.byref(n_out), C.byref(thr), None)
        assert rc == _ffi.PCRS_ERR_INVALID_ARG
        assert b"2^32-2" in lib.pcrs_last_error(ctx.handle)
        h = C.c_void_p()
        rc = lib.pcrs_index_build_dev(ctx.handle, p, p, p, n, 16, 0, C.byref(h))
        assert rc == _ffi.PCRS_ERR_INVALID_ARG and not h.value
        rc = lib.pcrs_voxel_downsample(ctx.handle, p, p, p, n, 1.0, p, p, p, C.byref(n_out))
        assert rc == _ffi.PCRS_ERR_INVALID_ARG
    # a device allocation the GPU cannot satisfy: out of memory, not an abort
    h = C.c_void_p()
    rc = lib.pcrs_index_build(ctx.handle, p, p, p, 16, 16, 0, C.byref(h))
    assert rc == _ffi.PCRS_OK
    big = (1 << 32) - 3
    idx = np.zeros(16, dtype=np.uint32)
    rc = lib.pcrs_knn_batch(h, p, p, p, 16, big, idx.ctypes.data_as(C.c_void_p), None, None)
    assert rc in (_ffi.PCRS_ERR_OOM, _ffi.PCRS_ERR_HIP, _ffi.PCRS_ERR_INVALID_ARG)
    assert len(lib.pcrs_last_error(ctx.handle)) > 0
    lib.pcrs_index_destroy(h)
    # the context is still usable afterwards
    from pointclouds_rs import hip, synth

    keep2, n_keep, _, _ = hip.sor(*synth.uniform_cloud(5000, 3), 8, 1.0)
    assert 0 < n_keep <= 5000


@pytest.mark.gpu
def test_results_are_identical_from_run_to_run():
    """The order of the points inside a grid cell comes from LDS atomics and differs between two
    builds of the same index; no output may depend on it."""
    from pointclouds_rs import hip, synth

    x, y, z = synth.uniform_cloud(300_000, 77, 0.0, 60.0)
    x, y, z = x.copy(), y.copy(), z.copy()
    x[1000:1200], y[1000:1200], z[1000:1200] = x[0:200], y[0:200], z[0:200]  # exact duplicates
    runs = []
    for _ in range(3):
        keep, n_keep, thr, nrm = hip.sor_normals(x, y, z, 16, 1.0, 16)
        ix = hip.Index(x, y, z, k_hint=10, algo=2)
        idx, dist, cnt = ix.knn(x[:50_000], y[:50_000], z[:50_000], 10)
        ix.close()
        runs.append((keep.tobytes(), n_keep, F(thr).tobytes(), nrm[0].tobytes(), nrm[1].tobytes(),
                     nrm[2].tobytes(), idx.tobytes(), dist.tobytes(), cnt.tobytes()))
    assert runs[0] == runs[1] == runs[2]
