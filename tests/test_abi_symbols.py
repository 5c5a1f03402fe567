"""The shared library must load and export every symbol include/pcrs_hip.h declares
(no compute is issued here, so this runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pcrs_hip.h")
LIB = os.path.join(ROOT, "pointclouds-rs_amd", "lib", "libpcrs_hip.so")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pcrs_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_expected_entry_points():
    names = declared_functions()
    for must in ("pcrs_ctx_create", "pcrs_index_build", "pcrs_knn_batch", "pcrs_sor",
                 "pcrs_normals", "pcrs_sor_normals", "pcrs_icp_run", "pcrs_apply_transform"):
        assert must in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build with `make -C pointclouds-rs_amd/csrc` or __graft_entry__.build()"
    lib = ctypes.CDLL(LIB)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, f"not exported: {missing}"


def test_python_binding_covers_header():
    from pointclouds_rs import _ffi

    assert sorted(_ffi.SIGNATURES) == declared_functions()
    _ffi.load_library()  # declares argtypes for every symbol


def test_version_and_pure_host_entry_points():
    from pointclouds_rs import _ffi, hip
    import numpy as np

    lib = _ffi.load_library()
    assert b"gfx950" in lib.pcrs_version()
    # pcrs_compose and pcrs_icp_solve are pure host math: callable without a GPU
    r, t = hip.compose(np.eye(3), (1, 0, 0), np.eye(3), (0, 2, 0))
    assert list(t) == [1, 2, 0] and np.array_equal(r, np.eye(3, dtype=np.float32))
    sums = np.zeros(_ffi.ICP_NSUMS)
    rot, tr = hip.icp_solve(_ffi.ICP_POINT_TO_PLANE, sums)  # no correspondences -> identity
    assert np.array_equal(rot, np.eye(3, dtype=np.float32)) and not tr.any()


def test_no_gpu_fails_loudly():
    """Without a GPU the product path must raise, never fall back to a CPU computation."""
    import subprocess, sys

    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from pointclouds_rs import _ffi\n"
        "import ctypes as C\n"
        "lib = _ffi.load_library(); h = C.c_void_p()\n"
        "rc = lib.pcrs_ctx_create(0, C.byref(h))\n"
        "print('rc', rc)\n" % os.path.join(ROOT, "pointclouds-rs_amd")
    )
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert "rc -2" in out.stdout or "rc -1" in out.stdout, out.stdout + out.stderr
