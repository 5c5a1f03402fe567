"""PCD reader / writer of the Python mirror against the reference's own io tests
(crates/io/src/pcd.rs:316-430) and the layout of its sample file (data/two_scans.pcd:1-13)."""
import numpy as np
import pytest

import pointclouds_rs as pc

F = np.float32

TWO_SCANS = (  # data/two_scans.pcd, restated (PCD v0.7 ASCII, FIELDS x y z)
    "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\n"
    "COUNT 1 1 1\nWIDTH 2\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 2\nDATA ascii\n0.0 0.0 0.0\n1.0 1.0 1.0\n"
)


def _cloud(x, y, z):
    return pc.PointCloud.from_numpy(np.stack([np.asarray(x, F), np.asarray(y, F), np.asarray(z, F)], 1))


def test_pcd_roundtrip(tmp_path):  # pcd.rs:316-331
    c = _cloud([1, 2, 3], [4, 5, 6], [7, 8, 9])
    p = tmp_path / "a.pcd"
    pc.write_pcd(str(p), c)
    assert np.array_equal(pc.read_pcd(str(p)).to_numpy(), c.to_numpy())
    assert p.read_text().endswith("DATA ascii\n1 4 7\n2 5 8\n3 6 9\n")  # Rust `{}` of 1.0f32 is "1"


def test_pcd_empty_cloud(tmp_path):  # pcd.rs:334-340, :371-377
    for writer in (pc.write_pcd, pc.write_pcd_binary):
        p = tmp_path / "e.pcd"
        writer(str(p), pc.PointCloud())
        assert pc.read_pcd(str(p)).is_empty()


def test_pcd_read_sample_layout(tmp_path):  # pcd.rs:343-351 / data/two_scans.pcd
    p = tmp_path / "two_scans.pcd"
    p.write_text(TWO_SCANS)
    c = pc.read_pcd(str(p))
    assert np.array_equal(c.to_numpy(), np.array([[0, 0, 0], [1, 1, 1]], F))


def test_pcd_binary_roundtrip(tmp_path):  # pcd.rs:354-368
    c = _cloud([1.5, -2.5, 3.0], [4.0, 5.25, -6.0], [7.0, 8.0, 9.125])
    p = tmp_path / "b.pcd"
    pc.write_pcd_binary(str(p), c)
    assert np.array_equal(pc.read_pcd(str(p)).to_numpy(), c.to_numpy())


def test_pcd_roundtrips_preserve_data(tmp_path):  # pcd.rs:379-430 (proptests, sampled)
    rng = np.random.default_rng(8)
    for n in (0, 1, 37, 200):
        a = rng.uniform(-1000, 1000, size=(n, 3)).astype(F)
        c = pc.PointCloud.from_numpy(a)
        for writer in (pc.write_pcd, pc.write_pcd_binary):
            p = tmp_path / "r.pcd"
            writer(str(p), c)
            got = pc.read_pcd(str(p)).to_numpy().reshape(-1, 3)
            assert np.array_equal(got.view(np.uint32), a.view(np.uint32))


def test_pcd_extra_fields_and_errors(tmp_path):
    # binary with an intensity field: x, y, z are picked by name (pcd.rs:271-285)
    hdr = ("VERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 2\n"
           "HEIGHT 1\nPOINTS 2\nDATA binary\n")
    body = np.array([[1, 2, 3, 9], [4, 5, 6, 9]], dtype="<f4").tobytes()
    p = tmp_path / "i.pcd"
    p.write_bytes(hdr.encode() + body)
    assert np.array_equal(pc.read_pcd(str(p)).to_numpy(), np.array([[1, 2, 3], [4, 5, 6]], F))
    p.write_bytes(hdr.encode() + body[:-4])  # pcd.rs:256-268
    with pytest.raises(IOError):
        pc.read_pcd(str(p))
    p.write_text("VERSION 0.7\nFIELDS x y z\nPOINTS 1\n")  # no DATA line, pcd.rs:107-110
    with pytest.raises(IOError):
        pc.read_pcd(str(p))
    p.write_text("FIELDS x y z\nPOINTS 1\nDATA binary_compressed\n")  # pcd.rs:137-140
    with pytest.raises(IOError):
        pc.read_pcd(str(p))
