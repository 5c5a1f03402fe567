"""Host-side mirror of the PyO3 surface: argument validation and cloud bookkeeping
(tests/test_python.py of the reference, the parts that need no device)."""
import numpy as np
import pytest

import pointclouds_rs as pc


def test_from_numpy_roundtrip_and_types():
    a = np.arange(12, dtype=np.float32).reshape(4, 3)
    c = pc.PointCloud.from_numpy(a)
    assert c.len() == 4 and len(c) == 4 and not c.is_empty()
    assert np.array_equal(c.to_numpy(), a)
    assert repr(c) == "PointCloud(n=4)"
    c64 = pc.PointCloud.from_numpy(a.astype(np.float64))
    assert c64.to_numpy().dtype == np.float32
    with pytest.raises(TypeError):
        pc.PointCloud.from_numpy(a.astype(np.int32))
    with pytest.raises(ValueError):
        pc.PointCloud.from_numpy(np.asfortranarray(a))
    with pytest.raises(ValueError):
        pc.PointCloud.from_numpy(np.zeros((4, 2), dtype=np.float32))


def test_select_and_inverse():
    a = np.arange(15, dtype=np.float32).reshape(5, 3)
    c = pc.PointCloud.from_numpy(a)
    assert np.array_equal(c.select([3, 1]).to_numpy(), a[[3, 1]])
    assert np.array_equal(c.select_inverse([0, 4]).to_numpy(), a[[1, 2, 3]])
    with pytest.raises(IndexError):
        c.select([5])
    with pytest.raises(IndexError):
        c.select_inverse([7])


def test_select_carries_every_attribute():
    # crates/core/src/cloud.rs:103-140: normals, colors and intensity follow the selected points
    a = np.arange(18, dtype=np.float32).reshape(6, 3)
    c = pc.PointCloud.from_numpy(a)
    c._normals = tuple(np.arange(6, dtype=np.float32) + 10 * k for k in range(3))
    c._colors = tuple((np.arange(6) + 40 * k).astype(np.uint8) for k in range(3))
    c._intensity = np.linspace(0.0, 1.0, 6, dtype=np.float32)
    s = c.select([4, 0, 4])
    assert np.array_equal(s.to_numpy(), a[[4, 0, 4]])
    assert all(np.array_equal(s._normals[k], c._normals[k][[4, 0, 4]]) for k in range(3))
    assert all(np.array_equal(s._colors[k], c._colors[k][[4, 0, 4]]) for k in range(3))
    assert s._colors[0].dtype == np.uint8 and np.array_equal(s._intensity, c._intensity[[4, 0, 4]])
    inv = c.select_inverse([1, 2])
    assert np.array_equal(inv._intensity, c._intensity[[0, 3, 4, 5]])
    assert np.array_equal(inv._colors[2], c._colors[2][[0, 3, 4, 5]])
    plain = pc.PointCloud.from_numpy(a).select([1])
    assert plain._normals is None and plain._colors is None and plain._intensity is None


def test_argument_validation_matches_reference():
    c = pc.PointCloud.from_numpy(np.zeros((2, 3), dtype=np.float32))
    for bad in (float("nan"), float("inf"), -1.0):
        with pytest.raises(ValueError):
            pc.statistical_outlier_removal(c, 5, bad)  # crates/python/src/filters.rs:42-46
    for bad in (0.0, -1.0, float("inf"), float("nan")):
        with pytest.raises(ValueError):
            pc.radius_outlier_removal(c, bad, 3)  # filters.rs:57-61
    with pytest.raises(ValueError):
        pc.icp_point_to_plane(c, c)  # registration.rs:67-72: target has no normals
    with pytest.raises(TypeError):
        pc.statistical_outlier_removal(np.zeros((2, 3)), 5, 1.0)
    # the PyO3 signature takes f32: a value beyond its range is infinite when it is checked
    # (filters.rs:5-10,40-46,54-61)
    with pytest.raises(ValueError):
        pc.statistical_outlier_removal(c, 5, 1e39)
    with pytest.raises(ValueError):
        pc.voxel_downsample(c, 1e39)
    with pytest.raises(ValueError):
        pc.radius_outlier_removal(c, 1e39, 3)
    with pytest.raises(ValueError):
        pc.voxel_downsample(c, 1e-60)  # rounds to 0.0f
    # unsigned arguments fail at conversion, before the body's checks and early returns
    empty = pc.PointCloud()
    with pytest.raises(OverflowError):
        pc.radius_outlier_removal(empty, 1.0, -1)
    with pytest.raises(OverflowError):
        pc.radius_outlier_removal(c, -1.0, -1)
    with pytest.raises(OverflowError):
        pc.statistical_outlier_removal(c, -1, float("nan"))


def test_out_of_scope_functions_are_absent():
    for name in ("passthrough_filter",):
        assert not hasattr(pc, name)


def test_sample_triples_vectorised_road_equals_the_definition():
    """All draws at once (SplitMix64 is a function of the draw counter) must give the triples of
    the draw-by-draw procedure, whether or not a redraw happens."""
    import pointclouds_rs as pc

    for n, it, seed in [(45000, 500, 1234), (100000, 1000, 1), (3, 50, 9), (10, 200, 77),
                        (5000, 2000, 2 ** 63 + 12345), (4_000_000_000, 100, 5), (70000, 0, 3),
                        (2, 10, 1)]:
        a = pc.sample_triples(n, it, seed)
        b = pc._sample_triples_scalar(n, it, seed) if n >= 3 else np.zeros((0, 3), np.uint32)
        assert a.dtype == np.uint32 and a.shape == b.shape and np.array_equal(a, b), (n, it, seed)


def test_sample_triples_of_the_library_are_the_mirrors():
    """pcrs_sample_triples (host code of the library, used inside pcrs_obstacle_pipeline) against
    the Python definition, draw for draw, redraws and dropped triples included."""
    import pointclouds_rs as pc
    from pointclouds_rs import hip

    for n, it, seed in ((1000, 500, 1234), (5, 60, 7), (3, 40, 99), (44000, 500, 2 ** 63 + 5), (2, 10, 1),
                        (4, 200, 0)):
        a = pc._sample_triples_scalar(n, it, seed)
        b = hip.sample_triples(n, it, seed)
        assert np.array_equal(a, b), (n, it, seed)
        assert np.array_equal(pc.sample_triples(n, it, seed), a)
