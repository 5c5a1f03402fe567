"""pcrs_sor_select_static_dev (the entry point that allocates nothing: its work space is a static
device array of the library) against pcrs_sor_select_dev, bit for bit, on two calls in a row on
the same GPU — the second on a smaller input, which must not see what the first left behind."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
F = np.float32


def _decode(result):
    raw = result.cpu().numpy().tobytes()
    thr = np.frombuffer(raw, dtype=np.float32, count=1, offset=0)[0]
    ok = np.frombuffer(raw, dtype=np.int32, count=1, offset=4)[0]
    nk = np.frombuffer(raw, dtype=np.uint64, count=1, offset=8)[0]
    return thr, bool(ok), int(nk)


def _both(engine, m, b, c, mul):
    d = engine.from_numpy(m)
    sub = d[b:b + c]
    keep_ref = engine.empty(c, "uint8")
    thr_ref, ok_ref, nk_ref = engine.select(d, mul, sub, keep_ref)
    keep = engine.empty(c, "uint8")
    keep.fill_(7)
    result = engine.torch.zeros(16, dtype=engine.torch.uint8, device=engine.device)
    engine.select_static(d, mul, sub, keep, result)
    engine.torch.cuda.synchronize()
    thr, ok, nk = _decode(result)
    assert ok == ok_ref and nk == nk_ref
    if ok:
        assert F(thr).view(np.uint32) == F(thr_ref).view(np.uint32), (thr, thr_ref)
    else:
        assert np.isnan(thr)
    assert np.array_equal(keep.cpu().numpy(), keep_ref.cpu().numpy())
    return thr, nk


def test_static_select_matches_select_twice_in_a_row():
    from pointclouds_rs.dist import HipEngine

    engine = HipEngine.shared(0)
    rng = np.random.default_rng(31)
    # first call: past 2^20 means, so the composites are built for four binades (largest layout)
    big = rng.lognormal(0.0, 1.0, 3_000_000).astype(F)
    big[::100003] = np.inf
    big[5::70001] = np.nan
    t1, n1 = _both(engine, big, 4096, 1_234_567, 1.0)
    # second call, smaller input, different layout of the same array (one binade per chunk)
    small = rng.uniform(0.2, 3.0, 70_001).astype(F)
    small[::997] = np.inf
    t2, n2 = _both(engine, small, 0, len(small), 2.5)
    assert t1 != t2 and n1 != n2
    # and a third, smaller again, with no finite mean at all: nothing kept, NaN threshold
    none = np.full(1000, np.inf, dtype=F)
    _both(engine, none, 0, 1000, 1.0)


def test_static_select_rejects_sizes_beyond_its_array():
    from pointclouds_rs import _ffi
    from pointclouds_rs.dist import HipEngine

    engine = HipEngine.shared(0)
    n = (1 << 22) + 1
    d = engine.torch.ones(n, dtype=engine.torch.float32, device=engine.device)
    keep = engine.empty(16, "uint8")
    result = engine.torch.zeros(16, dtype=engine.torch.uint8, device=engine.device)
    with pytest.raises(_ffi.PcrsError) as e:
        engine.select_static(d, 1.0, d[:16], keep, result)
    assert e.value.code == _ffi.PCRS_ERR_INVALID_ARG and "at most" in e.value.message
    # the largest size it accepts
    engine.select_static(d[:1 << 22], 1.0, d[:16], keep, result)
    engine.torch.cuda.synchronize()
    thr, ok, nk = _decode(result)
    assert ok and nk == 16 and thr == 1.0


def test_static_select_runs_under_stream_capture():
    """The property the entry point exists for: it can be recorded into a HIP graph.  A device
    allocation, a release or a wait for the stream inside the call would invalidate the capture;
    the replayed graph must then give what pcrs_sor_select_dev gives, twice, the second time on
    other values in the same buffers."""
    from pointclouds_rs.dist import HipEngine

    engine = HipEngine(0, use_torch_stream=False)  # a context of its own, bound to a side stream
    try:
        torch = engine.torch
        side = torch.cuda.Stream(device=engine.device)
        engine.ctx.set_stream(side.cuda_stream)
        rng = np.random.default_rng(5)
        n, b, c, mul = 600_000, 1000, 300_000, 1.5
        host = rng.lognormal(0.0, 0.7, n).astype(F)
        host[::50021] = np.inf
        d = engine.from_numpy(host)
        sub = d[b:b + c]
        keep = engine.empty(c, "uint8")
        result = torch.zeros(16, dtype=torch.uint8, device=engine.device)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            engine.select_static(d, mul, sub, keep, result)
        for rep in range(2):
            if rep == 1:  # new values in the captured buffers
                host = rng.uniform(0.1, 4.0, n).astype(F)
                host[7::9973] = np.nan
                d.copy_(torch.from_numpy(host))
            keep.fill_(7)
            result.zero_()
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            thr, ok, nk = _decode(result)
            keep_ref = engine.empty(c, "uint8")
            thr_ref, ok_ref, nk_ref = engine.select(d, mul, sub, keep_ref)
            torch.cuda.synchronize()
            assert ok and ok_ref and nk == nk_ref
            assert F(thr).view(np.uint32) == F(thr_ref).view(np.uint32), (thr, thr_ref)
            assert np.array_equal(keep.cpu().numpy(), keep_ref.cpu().numpy())
    finally:
        engine.close()
