"""The device-side SOR statistics must reproduce the reference's left-to-right f32 sums
(crates/filters/src/statistical_outlier.rs:43-62) bit for bit, for any data."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def engine():
    from pointclouds_rs.dist import HipEngine

    return HipEngine.shared(0)


def cases():
    rng = np.random.default_rng(5)
    out = {}
    for n in (1, 2, 63, 255, 256, 257, 1000, 5000, 70000, 1_000_000):
        out[f"uniform_{n}"] = rng.uniform(0.2, 3.0, n).astype(F)
    m = rng.uniform(0.5, 2.0, 300000).astype(F)
    m[::1000] = 5000.0  # heavy outliers: binade jumps in the middle of the stream
    out["outliers"] = m
    m = rng.uniform(0.5, 2.0, 100000).astype(F)
    m[::7] = np.inf
    m[3::11] = np.nan
    out["nonfinite"] = m
    out["all_inf"] = np.full(1000, np.inf, dtype=F)
    out["dyadic_ties"] = (rng.integers(0, 64, 200000) * 0.0625).astype(F)  # exact rounding ties
    out["constant_half"] = np.full(300000, 0.5, dtype=F)
    out["zeros_then_ones"] = np.concatenate([np.zeros(5000, F), np.ones(50000, F)])
    out["tiny"] = rng.uniform(1e-36, 1e-33, 20000).astype(F)  # below the scanned binades
    out["huge_range"] = np.exp(rng.uniform(-60, 60, 100000)).astype(F)
    out["descending"] = np.sort(rng.uniform(0.0, 10.0, 200000).astype(F))[::-1].copy()
    out["lognormal"] = rng.lognormal(0.0, 2.0, 400000).astype(F)
    return out


@pytest.mark.parametrize("name", list(cases()))
@pytest.mark.parametrize("mul", [0.0, 1.0, 2.5])
def test_threshold_bit_exact(engine, oracle, name, mul):
    m = cases()[name]
    ok_ref, thr_ref, mu_ref, sd_ref = oracle.sor_threshold(m, mul)
    d = engine.from_numpy(m)
    thr, ok = engine.threshold(d, mul)
    assert ok == ok_ref
    if ok:
        assert F(thr).view(np.uint32) == F(thr_ref).view(np.uint32), (name, thr, thr_ref)
    else:
        assert np.isnan(thr)


def test_mask_matches(engine, oracle):
    m = cases()["outliers"]
    _, thr_ref, _, _ = oracle.sor_threshold(m, 1.0)
    d = engine.from_numpy(m)
    thr, ok = engine.threshold(d, 1.0)
    keep = engine.empty(len(m), "uint8")
    n_keep = engine.mask(d, thr, keep)
    ref = m <= F(thr_ref)
    assert n_keep == int(ref.sum())
    assert np.array_equal(keep.cpu().numpy().astype(bool), ref)


@pytest.mark.parametrize("name", ["outliers", "lognormal", "tiny"] if "tiny" in cases() else ["outliers", "lognormal"])
def test_select_equals_threshold_then_mask(engine, oracle, name):
    """pcrs_sor_select_dev (threshold over all means, mask of a sub-range, one read-back) against
    the two-call path, including unaligned sub-ranges (the vectorised body needs 16-byte means)."""
    m = cases()[name]
    ok_ref, thr_ref, _, _ = oracle.sor_threshold(m, 1.0)
    d = engine.from_numpy(m)
    for b, c in ((0, len(m)), (1, 1001), (4096, 12345), (len(m) - 7, 7)):
        sub = d[b:b + c]
        keep = engine.empty(c + 3, "uint8")[3:] if b % 2 else engine.empty(c, "uint8")
        thr, ok, n_keep = engine.select(d, 1.0, sub, keep)
        assert ok == ok_ref and F(thr).view(np.uint32) == F(thr_ref).view(np.uint32)
        ref = m[b:b + c] <= F(thr_ref)
        assert n_keep == int(ref.sum())
        assert np.array_equal(keep.cpu().numpy().astype(bool), ref)
    nan = engine.from_numpy(np.full(1000, np.inf, dtype=F))
    keep = engine.empty(1000, "uint8")
    keep.fill_(1)
    thr, ok, n_keep = engine.select(nan, 1.0, nan, keep)
    assert not ok and np.isnan(thr) and n_keep == 0 and not keep.cpu().numpy().any()


@pytest.mark.parametrize("name", ["stall20M", "lognormal6M", "tiny5M", "mixed4M", "mixed1.1M",
                                  "ascending9M", "mixed1M"])
def test_threshold_bit_exact_long_sums(engine, oracle, name):
    """Sums long enough for the sequential f32 sum to fall behind the exact sum (it stalls once
    the elements are below half an ulp): the composites are then built for a window of binades."""
    rng = np.random.default_rng(12)
    if name == "stall20M":
        m = rng.uniform(0.5, 1.5, 20_000_000).astype(F)
    elif name == "lognormal6M":
        m = rng.lognormal(0.0, 1.5, 6_000_000).astype(F)
    elif name == "tiny5M":
        m = (rng.uniform(0.0, 1.0, 5_000_000) * 1e-12).astype(F)
    elif name.startswith("mixed"):
        # three quarters of the values identical: their squared deviations are far below half an
        # ulp of the variance sum and all round the same way (the largest drift the sum allows)
        n = {"mixed4M": 4_000_000, "mixed1.1M": 1_100_000, "mixed1M": 1_000_000}[name]
        m = rng.gamma(9.0, 0.13, n).astype(F)
        m[rng.random(n) < 0.75] = 1.0
    else:  # sorted input, just past the unanchored first section of 2^23 elements
        m = np.sort(rng.gamma(2.0, 0.5, 9_000_000).astype(F))
    m[::100003] = np.inf
    for mul in (1.0,):
        ok_ref, thr_ref, _, _ = oracle.sor_threshold(m, mul)
        thr, ok = engine.threshold(engine.from_numpy(m), mul)
        assert ok == ok_ref and F(thr).view(np.uint32) == F(thr_ref).view(np.uint32), (name, thr, thr_ref)
