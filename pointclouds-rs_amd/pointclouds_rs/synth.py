"""Deterministic synthetic clouds for the tests and the benchmark (shapes follow the reference's
benches and integration tests; the PRNG is SplitMix64 so the data does not depend on a library
version — the reference's `StdRng` stream cannot be reproduced without the `rand` crate)."""
import numpy as np

F = np.float32
_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n, stream=0):
    """n uint64 values: element i is SplitMix64's (i+1)-th output for state seed + stream * 2^32."""
    with np.errstate(over="ignore"):
        i = np.arange(1, n + 1, dtype=np.uint64)
        s = np.uint64(seed) + np.uint64(stream) * np.uint64(1 << 32)
        z = s + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed, n, stream=0):
    """float32 in [0, 1) with 24 random bits."""
    return ((splitmix64(seed, n, stream) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))).astype(F)


def uniform(seed, n, lo, hi, stream=0):
    return (F(lo) + uniform01(seed, n, stream) * F(hi - lo)).astype(F)


def uniform_cloud(n, seed=42, lo=0.0, hi=100.0):
    """benches/bench_filters.rs:9-15 shape: n points uniform in [lo, hi)^3."""
    return (uniform(seed, n, lo, hi, 0), uniform(seed, n, lo, hi, 1), uniform(seed, n, lo, hi, 2))


def hemisphere(n, seed, radius):
    """tests/real_world_pipeline.rs:58-80: rejection-sampled upper hemisphere of the given radius."""
    xs, ys = [], []
    got, stream = 0, 0
    while got < n:
        m = max(2 * (n - got), 64)
        px = uniform(seed, m, -1.0, 1.0, 10 + 2 * stream)
        py = uniform(seed, m, -1.0, 1.0, 11 + 2 * stream)
        ok = (px * px + py * py) < F(1.0)
        xs.append(px[ok])
        ys.append(py[ok])
        got += int(ok.sum())
        stream += 1
    px = np.concatenate(xs)[:n]
    py = np.concatenate(ys)[:n]
    pz = np.sqrt(F(1.0) - (px * px + py * py)).astype(F)
    r = F(radius)
    return (px * r).astype(F), (py * r).astype(F), (pz * r).astype(F)


def kitti_scene(seed=42, ground=50_000, per_obstacle=2_000, outliers=1_000):
    """tests/real_world_pipeline.rs:19-55: ground plane + 3 box obstacles + scattered outliers."""
    parts = []
    parts.append((uniform(seed, ground, -20, 20, 0), uniform(seed, ground, -20, 20, 1),
                  uniform(seed, ground, -0.05, 0.05, 2)))
    centers = [(5.0, 3.0, 1.0), (-8.0, -5.0, 0.8), (12.0, -10.0, 1.5)]
    for i, c in enumerate(centers):
        s = 3 + 3 * i
        parts.append((F(c[0]) + uniform(seed, per_obstacle, -0.8, 0.8, s),
                      F(c[1]) + uniform(seed, per_obstacle, -0.8, 0.8, s + 1),
                      F(c[2]) + uniform(seed, per_obstacle, -0.5, 0.5, s + 2)))
    parts.append((uniform(seed, outliers, -30, 30, 20), uniform(seed, outliers, -30, 30, 21),
                  uniform(seed, outliers, -5, 10, 22)))
    return tuple(np.concatenate([p[a] for p in parts]).astype(F) for a in range(3))


def rot_z(angle):
    a = F(angle)
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=F)


def degenerate_cloud(n=300, seed=7):
    """Duplicates, a lattice full of distance ties, co-planar points and non-finite rows."""
    x, y, z = uniform_cloud(n, seed, 0.0, 4.0)
    x, y, z = x.copy(), y.copy(), z.copy()
    # exact duplicates
    x[10:20], y[10:20], z[10:20] = x[0:10], y[0:10], z[0:10]
    # integer lattice block (many equidistant neighbours)
    g = np.array([(i, j, k) for i in range(4) for j in range(4) for k in range(4)], dtype=F)
    x[40:104], y[40:104], z[40:104] = g[:, 0], g[:, 1], g[:, 2]
    # co-planar patch (would overflow a kiddo bucket on one axis)
    z[120:200] = F(2.5)
    # non-finite rows
    x[5] = np.nan
    y[25] = np.inf
    z[33] = -np.inf
    return x, y, z


def ring_scan(sweeps=1, seed=11, beams=64, azimuths=2048, height=1.73):
    """A spinning-LiDAR shaped scan of a ground plane: `beams` rings of `azimuths` returns per
    sweep around a sensor `height` above the ground, several sweeps of a moving sensor.  Dense
    rings near the sensor, sparse ones far out: the density falls with the square of the range
    (3.7 m to 99 m), by a factor of several hundred over the scan."""
    rng = np.random.default_rng(seed)
    elev = np.deg2rad(np.linspace(-24.8, -1.0, beams))  # below the horizon: ground hits
    r = height / np.tan(-elev)
    parts = []
    for s in range(sweeps):
        a = np.linspace(0, 2 * np.pi, azimuths, endpoint=False) + rng.uniform(0, 2 * np.pi / azimuths)
        rr = (r[:, None] * (1 + rng.normal(0, 2e-3, (beams, azimuths)))).ravel()
        aa = np.broadcast_to(a, (beams, azimuths)).ravel()
        parts.append(np.stack([rr * np.cos(aa) + 0.8 * s, rr * np.sin(aa),
                               rng.normal(0, 0.02, rr.size)], 1))
    p = np.concatenate(parts).astype(F)
    return tuple(np.ascontiguousarray(p[:, i]) for i in range(3))


def blob_in_box(n_blob, n_background, seed=0, sigma=1e-3, centre=3.0, half=100.0):
    """A tight Gaussian blob (a few millimetres across) inside a sparse uniform box 200 units wide,
    shuffled: the whole blob falls into one cell of any grid sized for the box."""
    rng = np.random.default_rng(seed)
    c = (rng.normal(0, sigma, size=(n_blob, 3)) + centre).astype(F)
    bg = rng.uniform(-half, half, size=(n_background, 3)).astype(F)
    p = np.concatenate([c, bg])
    p = p[rng.permutation(len(p))]
    return tuple(np.ascontiguousarray(p[:, i]) for i in range(3))


def clusters_in_field(n_each=40_000, n_background=150_000, seed=5, sigma=0.3, half=50.0):
    """Four dense Gaussian clusters (a fraction of a unit across) far apart in a sparse uniform box
    100 units wide, shuffled: crowded cells in four corners of any grid sized for the box, so that
    one box around them is most of the grid."""
    rng = np.random.default_rng(seed)
    centres = np.array([[-30, -30, -30], [30, -30, -20], [-30, 30, 20], [25, 25, 30]], dtype=np.float64)
    parts = [rng.normal(0, sigma, size=(n_each, 3)) + c for c in centres]
    parts.append(rng.uniform(-half, half, size=(n_background, 3)))
    p = np.concatenate(parts).astype(F)
    p = p[rng.permutation(len(p))]
    return tuple(np.ascontiguousarray(p[:, i]) for i in range(3))


def _normal(seed, n, sigma, stream):
    """Normal deviates (Box-Muller over two SplitMix64 streams), float32."""
    u1 = (splitmix64(seed, n, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    u2 = (splitmix64(seed, n, stream + 1) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return (sigma * np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)).astype(F)


def aerial_scene(scale=1.0, seed=42, with_parts=False):
    """The scene of examples/python/aerial_lidar.py:26-137: a 500 m x 500 m survey area with an
    undulating terrain surface (2 M points x scale, 5 cm noise), five flat-roofed buildings (50 K
    points x scale each: 70 % roof, 30 % on the four walls) and eight trees (20 K points x scale
    each: 80 % on a hemispherical canopy 4 m above the ground, 20 % along the trunk).  About
    2.41 M points at scale 1; points come part by part, as in the reference's scene.
    with_parts: also returns [(name, begin, end)] of the parts ("terrain", "roof", "wall",
    "canopy", "trunk").  The walls are exact planes x = const or y = const."""
    xs, ys, zs, parts = [], [], [], []

    def add(x, y, z, name):
        begin = sum(len(a) for a in xs)
        xs.append(np.asarray(x, dtype=F))
        ys.append(np.asarray(y, dtype=F))
        zs.append(np.asarray(z, dtype=F))
        parts.append((name, begin, begin + len(xs[-1])))

    st = 0
    n = int(2_000_000 * scale)
    tx, ty = uniform(seed, n, 0.0, 500.0, st), uniform(seed, n, 0.0, 500.0, st + 1)
    tz = (2.0 * np.sin(tx.astype(np.float64) * 0.02) * np.cos(ty.astype(np.float64) * 0.015)).astype(F)
    add(tx, ty, tz + _normal(seed, n, 0.05, st + 2), "terrain")
    st += 4
    buildings = ((100, 120, 30, 20, 12), (250, 300, 40, 40, 18), (350, 100, 25, 25, 8),
                 (400, 400, 50, 30, 15), (150, 350, 20, 20, 10))
    per = int(50_000 * scale)
    for cx, cy, w, d, h in buildings:
        nr = int(per * 0.7)
        add(uniform(seed, nr, cx - w / 2, cx + w / 2, st), uniform(seed, nr, cy - d / 2, cy + d / 2, st + 1),
            F(h) + _normal(seed, nr, 0.02, st + 2), "roof")
        nw = per - nr
        side = (splitmix64(seed, nw, st + 4) >> np.uint64(62)).astype(np.int64)  # 0..3
        along_x = uniform(seed, nw, cx - w / 2, cx + w / 2, st + 5)
        along_y = uniform(seed, nw, cy - d / 2, cy + d / 2, st + 6)
        wx = np.where(side < 2, along_x, np.where(side == 2, F(cx + w / 2), F(cx - w / 2)))
        wy = np.where(side == 0, F(cy + d / 2), np.where(side == 1, F(cy - d / 2), along_y))
        add(wx, wy, uniform(seed, nw, 0.0, h, st + 7), "wall")
        st += 8
    trees = ((50, 50, 5), (180, 200, 6), (300, 450, 4), (420, 250, 7), (80, 400, 5), (450, 50, 4),
             (200, 80, 5), (350, 350, 6))
    per = int(20_000 * scale)
    for i, (cx, cy, r) in enumerate(trees):
        nc = int(per * 0.8)
        hx, hy, hz = hemisphere(nc, seed + 1000 + i, 1.0)
        add(F(cx) + hx * F(r), F(cy) + hy * F(r), F(4.0) + hz * F(r), "canopy")
        nt = per - nc
        add(F(cx) + _normal(seed, nt, 0.15, st), F(cy) + _normal(seed, nt, 0.15, st + 2),
            uniform(seed, nt, 0.0, 4.0, st + 4), "trunk")
        st += 5
    cols = (np.concatenate(xs), np.concatenate(ys), np.concatenate(zs))
    return cols + (parts,) if with_parts else cols
