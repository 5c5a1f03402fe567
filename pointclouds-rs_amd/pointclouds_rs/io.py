"""PCD files (x, y, z), following crates/io/src/pcd.rs of the reference: `read_pcd` (:7-21, ASCII
:202-236 and binary :238-309 bodies), `write_pcd` (:24-43), `write_pcd_binary` (:46-71).  Host-side
file handling only; the points end up in the same SoA f32 columns the GPU entry points take."""
from __future__ import annotations

import numpy as np


def _header_and_body(raw: bytes):
    """pcd.rs:82-127: the header runs up to and including the DATA line."""
    pos = 0
    while pos < len(raw):
        end = raw.find(b"\n", pos)
        line_end = len(raw) if end < 0 else end
        line = raw[pos:line_end]
        if line.lstrip().startswith(b"DATA"):
            return raw[:line_end].decode("utf-8", errors="strict"), (len(raw) if end < 0 else end + 1)
        if end < 0:
            break
        pos = end + 1
    raise ValueError("PCD file missing DATA line")


def _header_value(header: str, key: str):
    for line in header.splitlines():
        t = line.strip()
        if t.startswith(key):
            parts = t.split()
            if len(parts) >= 2:
                return parts[1:]
    return None


def _parse_f32(tok: str) -> np.float32:
    try:
        return np.float32(float(tok))  # pcd.rs:226-228: unparsable numbers read as 0
    except ValueError:
        return np.float32(0.0)


def read_pcd_columns(path):
    """Returns (x, y, z) float32 columns."""
    with open(path, "rb") as f:
        raw = f.read()
    try:
        header, body_at = _header_and_body(raw)
    except UnicodeDecodeError:
        raise ValueError("PCD header is not valid UTF-8")
    fmt = _header_value(header, "DATA")
    if not fmt:
        raise ValueError("PCD file missing DATA line")
    if fmt[0] not in ("ascii", "binary"):
        raise ValueError("unsupported PCD DATA format: %s" % fmt[0])
    fields = _header_value(header, "FIELDS") or ["x", "y", "z"]
    if fmt[0] == "ascii":
        rows = []
        for line in raw[body_at:].decode("utf-8").splitlines():
            t = line.strip()
            if not t or t.startswith("#"):
                continue
            parts = t.split()
            if len(parts) < 3:
                continue  # pcd.rs:222-224
            rows.append((_parse_f32(parts[0]), _parse_f32(parts[1]), _parse_f32(parts[2])))
        a = np.array(rows, dtype=np.float32).reshape(-1, 3)
        return a[:, 0].copy(), a[:, 1].copy(), a[:, 2].copy()
    count = _header_value(header, "POINTS") or _header_value(header, "WIDTH")
    if not count:
        raise ValueError("PCD file missing POINTS/WIDTH header")
    try:
        n = int(count[0])
        if n < 0:
            raise ValueError
    except ValueError:
        raise ValueError("invalid POINTS value: %s" % count[0])
    nf = len(fields)
    need = n * nf * 4  # pcd.rs:252-268: every field is read as a 4-byte float
    body = raw[body_at:]
    if len(body) < need:
        raise ValueError("binary PCD data too short: have %d bytes, expected %d (%d points x %d fields x 4)"
                         % (len(body), need, n, nf))
    try:
        ix, iy, iz = fields.index("x"), fields.index("y"), fields.index("z")
    except ValueError:
        raise ValueError("binary PCD file missing x, y, z fields")
    a = np.frombuffer(body, dtype="<f4", count=n * nf).reshape(n, nf)
    return a[:, ix].astype(np.float32), a[:, iy].astype(np.float32), a[:, iz].astype(np.float32)


def _header(n: int, data: str) -> str:
    return ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\n"
            "TYPE F F F\nCOUNT 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n"
            % (n, n, data))


def _fmt(v: np.float32) -> str:
    """Rust `{}` of an f32: the shortest decimal that reads back as the same f32, no exponent."""
    if np.isnan(v):
        return "NaN"
    if np.isinf(v):
        return "inf" if v > 0 else "-inf"
    return np.format_float_positional(np.float32(v), unique=True, trim="-")


def write_pcd_columns(path, x, y, z):
    n = len(x)
    with open(path, "w") as f:
        f.write(_header(n, "ascii"))
        for i in range(n):
            f.write("%s %s %s\n" % (_fmt(x[i]), _fmt(y[i]), _fmt(z[i])))


def write_pcd_binary_columns(path, x, y, z):
    n = len(x)
    a = np.stack([np.asarray(c, dtype="<f4") for c in (x, y, z)], axis=1) if n else np.zeros((0, 3), "<f4")
    with open(path, "wb") as f:
        f.write(_header(n, "binary").encode())
        f.write(np.ascontiguousarray(a).tobytes())
