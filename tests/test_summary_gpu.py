"""The end-of-run summary as HIP passes (fiveeq_row_moments_*, fiveeq_hist_rows_ranged_*, fiveeq_select_rows_*; host side
fiveeqscm_amd/distributed.py): moments and EXACT percentiles of device rows against NumPy, on the shapes that break
selection schemes — ties, constant rows, heavy tails, NaNs, rows of a handful of members, unaligned rows."""
import ctypes

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

PCT = (0.0, 5.0, 33.3, 49.999, 50.0, 50.001, 95.0, 100.0)


def _rows(n, rng):
    x = np.stack([
        rng.normal(1.8, 0.6, size=n),                                # an ensemble temperature row
        rng.uniform(size=n) ** 3,                                    # skewed towards the lower edge
        np.full(n, 2.5),                                             # constant
        np.round(rng.normal(size=n), 1),                             # ~80 distinct values: every bin is a heavy tie
        rng.standard_cauchy(size=n),                                 # heavy tails: nearly all members in a few of the 4096 bins
        np.where(rng.uniform(size=n) < 0.9, 0.25, rng.normal(size=n)),   # 90 % of the members share one value
    ])
    return x


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_percentile_selection_on_device_rows(dtype):
    from fiveeqscm_amd.distributed import device_row_sums, exact_percentiles, gather_summary
    rng = np.random.default_rng(2026)
    np_dt, t_dt = (np.float64, torch.float64) if dtype == "f64" else (np.float32, torch.float32)
    for n in (1, 2, 5, 63, 64, 65, 1000, 4097, 200_001, 3_000_001):
        xs = _rows(n, rng).astype(np_dt)
        t = torch.from_numpy(xs).cuda()
        st = {}
        out = gather_summary(t, PCT, stats=st)
        x64 = xs.astype(np.float64)
        want = np.percentile(x64, PCT, axis=1).T
        assert np.array_equal(out["percentiles"].cpu().numpy(), want), f"n={n}"          # np.percentile BIT FOR BIT
        assert st["bytes_to_root"] == 0 and st["allreduce_bytes"] == 0
        np.testing.assert_allclose(out["mean"].cpu().numpy(), x64.mean(1), rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(out["var"].cpu().numpy(), x64.var(1), rtol=1e-8, atol=1e-9 * (x64 ** 2).mean(1).max())
        assert np.array_equal(out["min"].cpu().numpy(), x64.min(1)) and np.array_equal(out["max"].cpu().numpy(), x64.max(1))
        assert out["count"].tolist() == [float(n)] * xs.shape[0]
        # the same selection with the extrema handed in (exact_percentiles' signature: no moments pass)
        sel = exact_percentiles(t, PCT, torch.from_numpy(x64.min(1)), torch.from_numpy(x64.max(1)), n)
        assert np.array_equal(sel.cpu().numpy(), want)
        # the moments pass is deterministic
        assert torch.equal(device_row_sums(t), device_row_sums(t))


def test_rows_with_a_nan_and_rows_that_are_not_aligned():
    from fiveeqscm_amd.distributed import device_row_sums, gather_summary
    rng = np.random.default_rng(7)
    n = 100_003
    x = rng.normal(size=(3, n))
    x[1, 77] = np.nan
    out = gather_summary(torch.from_numpy(x).cuda(), (5.0, 50.0, 95.0))
    pct = out["percentiles"].cpu().numpy()
    np.testing.assert_allclose(pct[[0, 2]], np.percentile(x[[0, 2]], (5.0, 50.0, 95.0), axis=1).T, rtol=1e-14)
    assert np.isnan(pct[1]).all() and np.isnan(out["mean"][1].item())            # like np.percentile / np.mean
    assert out["min"][1].item() == np.nanmin(x[1]) and out["max"][1].item() == np.nanmax(x[1])     # extrema skip the NaN
    # a view that starts one element into a wider buffer: rows neither 16-byte aligned nor contiguous (ld = n + 5)
    for dt, np_dt in ((torch.float64, np.float64), (torch.float32, np.float32)):
        wide = torch.from_numpy(rng.normal(size=(4, n + 5)).astype(np_dt)).cuda()
        view = wide[:, 1:n + 1]
        assert view.stride(0) == n + 5 and view.data_ptr() % 16 != 0
        got = device_row_sums(view).cpu().numpy()
        v = view.cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(got[:, 0], v.sum(1), rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(got[:, 1], (v * v).sum(1), rtol=1e-12)
        assert np.array_equal(got[:, 2], v.min(1)) and np.array_equal(got[:, 3], v.max(1))


def test_selection_and_pick_through_the_c_abi():
    """fiveeq_hist_rows_ranged_* / fiveeq_select_bins_* / fiveeq_select_pick_* called directly: the candidates are exactly the
    members the histogram counted in the marked bins (same rule, same ranges), a cap too small for them still counts them
    all, and the pick pass returns the k-th smallest candidate for every k asked — over one segment and over three."""
    from fiveeqscm_amd import _capi
    lib = _capi.load()
    rng = np.random.default_rng(3)
    n, nb = 300_007, 1000
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    for np_dt, sfx in ((np.float64, "f64"), (np.float32, "f32")):
        x = rng.normal(size=(2, n)).astype(np_dt)
        x[0, 5] = np.nan
        rows = torch.from_numpy(x).cuda()
        ranges = torch.tensor([[-1.5, 2.0], [np.nanmin(x[1]), np.nanmax(x[1])]], dtype=torch.float64, device="cuda")
        hist = torch.zeros((2, nb), dtype=torch.int64, device="cuda")
        _capi.check(lib, getattr(lib, f"fiveeq_hist_rows_ranged_{sfx}")(2, n, n, p(rows), p(ranges), nb, p(hist), None))
        counts = hist.cpu().numpy()
        assert counts.sum(1).tolist() == [n - 1, n]                                  # the NaN has no bin
        marked = rng.uniform(size=(2, nb)) < 0.01
        marked[:, [0, nb - 1]] = True                                                # the edge bins hold the outliers
        words = (nb + 31) // 32
        bits = np.zeros((2, words * 32), dtype=np.uint8)
        bits[:, :nb] = marked
        mask = torch.from_numpy(np.packbits(bits.reshape(2, words, 32), axis=2, bitorder="little").view(np.uint32).reshape(2, words).view(np.int32)).cuda()
        want_n = (counts * marked).sum(1)
        for cap in (100, int(want_n.max())):                                         # too small first: everything is still counted
            cand = torch.full((2, cap), -777.0, dtype=rows.dtype, device="cuda")
            cand_n = torch.zeros(2, dtype=torch.int64, device="cuda")
            _capi.check(lib, getattr(lib, f"fiveeq_select_bins_{sfx}")(2, n, n, p(rows), p(ranges), nb, p(mask), p(cand), cap,
                                                                       p(cand_n), None))
            assert cand_n.cpu().tolist() == want_n.tolist()
        got = [np.sort(cand[k, :want_n[k]].cpu().numpy()) for k in range(2)]
        # the candidates are whole bins: contiguous runs of the sorted row
        srt = np.sort(x[1])
        edges = np.concatenate([[0], np.cumsum(counts[1])])
        assert np.array_equal(got[1], np.concatenate([srt[edges[b]:edges[b + 1]] for b in np.nonzero(marked[1])[0]]))
        # pick: ranks 0, 1, middle, last, one past the end (-> NaN), negative (-> NaN)
        ranks = np.stack([[0, 1, want_n[k] // 2, want_n[k] - 1, want_n[k], -1] for k in range(2)]).astype(np.int64)
        picked = torch.zeros((2, 6), dtype=torch.float64, device="cuda")
        ranks_t = torch.from_numpy(ranks).cuda()                  # (kept alive: the calls below only see pointers)
        _capi.check(lib, getattr(lib, f"fiveeq_select_pick_{sfx}")(2, 1, cand.shape[1], p(cand), p(cand_n), 6, p(ranks_t),
                                                                   p(picked), None))
        pk = picked.cpu().numpy()
        for k in range(2):
            assert np.array_equal(pk[k, :4], got[k][ranks[k, :4]].astype(np.float64)) and np.isnan(pk[k, 4:]).all()
        # three segments of unequal length (the root of a three-rank exchange)
        width = int(want_n.max())
        pool = torch.full((2, 3, width), 9e9, dtype=rows.dtype, device="cuda")
        seg_n = np.zeros((2, 3), dtype=np.int64)
        for k in range(2):
            cuts = [0, want_n[k] // 5, want_n[k] // 2, want_n[k]]
            for g in range(3):
                seg_n[k, g] = cuts[g + 1] - cuts[g]
                pool[k, g, :seg_n[k, g]] = cand[k, cuts[g]:cuts[g + 1]]
        seg_t = torch.from_numpy(seg_n).cuda()
        _capi.check(lib, getattr(lib, f"fiveeq_select_pick_{sfx}")(2, 3, width, p(pool), p(seg_t), 6, p(ranks_t), p(picked), None))
        assert np.array_equal(picked.cpu().numpy()[:, :4], pk[:, :4])
    # argument checks on the host
    z = ctypes.c_void_p(0x1000)
    assert lib.fiveeq_select_bins_f64(1, 10, 10, z, z, 5000, z, z, 4, z, None) == _capi.E_INVALID
    assert lib.fiveeq_select_bins_f64(1, 10, 5, z, z, 64, z, z, 4, z, None) == _capi.E_INVALID
    assert lib.fiveeq_select_pick_f64(1, 0, 4, z, z, 2, z, z, None) == _capi.E_INVALID
    assert lib.fiveeq_row_moments_f64(1, 0, 0, z, z, z, None) == _capi.E_INVALID


def test_engine_summary_takes_its_moments_from_the_kernel_records():
    from fiveeqscm_amd import emissions, params
    from fiveeqscm_amd.engine import EnsembleEngine
    N, steps = 70_001, [20, 59]
    p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0")
    E = emissions.rcp_like_emissions(750, 3)[200:260]
    a = EnsembleEngine(p, N, E, device="cuda:0", output_steps=steps, store_concentrations=False, collect_stats=True)
    a.run(mode="fused")
    b = EnsembleEngine(p, N, E, device="cuda:0", output_steps=steps, store_concentrations=False)
    b.run()
    sa, sb = a.gather_summary(steps), b.gather_summary(steps)
    T = b.T.cpu().numpy()
    assert torch.equal(sa["percentiles"], sb["percentiles"])
    np.testing.assert_allclose(sa["percentiles"].cpu().numpy(), np.percentile(T, (5.0, 50.0, 95.0), axis=1).T, rtol=1e-14)
    for key in ("mean", "var", "min", "max"):
        np.testing.assert_allclose(sa[key].cpu().numpy(), sb[key].cpu().numpy(), rtol=1e-9 if key == "var" else 1e-13)
    np.testing.assert_allclose(sa["mean"].cpu().numpy(), T.mean(1), rtol=1e-13)
    with pytest.raises(ValueError, match="not stored"):
        a.gather_summary([3])
    # an engine that holds no moments for a step (a state-only checkpoint loaded, the step's row restored by hand) falls back
    # to the moments pass instead of reading zero-filled records
    c = EnsembleEngine(p, N, E, device="cuda:0", output_steps=steps, store_concentrations=False, collect_stats=True)
    c.load_state_dict(a.state_dict(include_outputs=False))
    c.T.copy_(a.T)
    assert not c._stats_have.any()
    sc = c.gather_summary(steps)
    assert torch.equal(sc["percentiles"], sb["percentiles"]) and torch.allclose(sc["mean"], sb["mean"], rtol=1e-13)


def test_engine_summary_of_a_gas_concentration():
    """"summary statistics of T (and optionally C)": gas= summarises that gas's stored C rows through the
    same four passes — np.percentile bit for bit, moments to rounding; T's summary is unchanged beside it."""
    from fiveeqscm_amd import emissions, params
    from fiveeqscm_amd.engine import EnsembleEngine
    N, steps = 30_011, [10, 59]
    p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0")
    E = emissions.rcp_like_emissions(750, 3)[200:260]
    eng = EnsembleEngine(p, N, E, device="cuda:0", output_steps=steps, collect_stats=True)
    eng.run(mode="fused")
    C = eng.C.cpu().numpy()
    for g in range(3):
        s = eng.gather_summary(steps, percentiles=PCT, gas=g)
        assert np.array_equal(s["percentiles"].cpu().numpy(), np.percentile(C[:, g], PCT, axis=1).T), g
        np.testing.assert_allclose(s["mean"].cpu().numpy(), C[:, g].mean(1), rtol=1e-13)
        assert np.array_equal(s["min"].cpu().numpy(), C[:, g].min(1)) and np.array_equal(s["max"].cpu().numpy(), C[:, g].max(1))
    assert np.array_equal(eng.gather_summary(steps, percentiles=PCT)["percentiles"].cpu().numpy(),
                          np.percentile(eng.T.cpu().numpy(), PCT, axis=1).T)
    with pytest.raises(ValueError, match="gases 0..2"):
        eng.gather_summary(steps, gas=3)
    no_c = EnsembleEngine(p, N, E, device="cuda:0", output_steps=steps, store_concentrations=False)
    with pytest.raises(RuntimeError, match="no stored C rows"):
        no_c.gather_summary(steps, gas=0)


def test_hip_passes_against_their_numpy_restatement():
    """oracle/summary_passes.py restates the four passes in NumPy behind the C ABI's signatures (it stands in for the library
    in the CPU tests of the multi-rank exchange).  Here the two meet on the same rows: moments, histogram counts, candidate
    sets and picked order statistics of the HIP kernels against the restatement — fp64 exactly; fp32 up to the restatement's
    double rounding of the fp32 FMA (a member within 2^-24 of a bin edge may sit in the neighbouring bin)."""
    from fiveeqscm_amd import _capi
    from oracle.summary_passes import SummaryPasses
    lib, cpu = _capi.load(), SummaryPasses()
    rng = np.random.default_rng(12)
    n, nb, K = 200_003, 4096, 3
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    for np_dt, sfx in ((np.float64, "f64"), (np.float32, "f32")):
        x = (rng.normal(1.8, 0.6, size=(K, n)) * np.array([[1.0], [0.3], [2.5]])).astype(np_dt)
        x[2, 100] = np.nan
        rows_d, rows_h = torch.from_numpy(x).cuda(), torch.from_numpy(x.copy())
        ranges = np.stack([np.nanmin(x, axis=1), np.nanmax(x, axis=1)], axis=1).astype(np.float64)
        rg_d, rg_h = torch.from_numpy(ranges).cuda(), torch.from_numpy(ranges.copy())
        # moments
        chunks = int(lib.fiveeq_row_moments_chunks(K, n))
        part = torch.empty(K * chunks * 4, dtype=torch.float64, device="cuda")
        mom_d, mom_h = torch.empty((K, 4), dtype=torch.float64, device="cuda"), torch.empty((K, 4), dtype=torch.float64)
        _capi.check(lib, getattr(lib, f"fiveeq_row_moments_{sfx}")(K, n, n, p(rows_d), p(part), p(mom_d), None))
        assert getattr(cpu, f"fiveeq_row_moments_{sfx}")(K, n, n, p(rows_h), None, p(mom_h), None) == 0
        a, b = mom_d.cpu().numpy(), mom_h.numpy()
        np.testing.assert_allclose(a[:2, :2], b[:2, :2], rtol=1e-12)
        assert np.array_equal(a[:, 2:], b[:, 2:]) and np.isnan(a[2, 0]) and np.isnan(b[2, 0])
        # histogram
        h_d, h_h = torch.zeros((K, nb), dtype=torch.int64, device="cuda"), torch.zeros((K, nb), dtype=torch.int64)
        _capi.check(lib, getattr(lib, f"fiveeq_hist_rows_ranged_{sfx}")(K, n, n, p(rows_d), p(rg_d), nb, p(h_d), None))
        assert getattr(cpu, f"fiveeq_hist_rows_ranged_{sfx}")(K, n, n, p(rows_h), p(rg_h), nb, p(h_h), None) == 0
        hd, hh = h_d.cpu().numpy(), h_h.numpy()
        assert hd.sum(1).tolist() == hh.sum(1).tolist() == [n, n, n - 1]
        moved = np.abs(hd - hh).sum(1) // 2
        assert (moved == 0).all() if sfx == "f64" else (moved <= 2).all(), moved
        # selection + pick on the bins that hold p05 / p50 / p95 of the device's own histogram
        cdf = np.cumsum(hd, axis=1)
        want = (np.array([0.05, 0.5, 0.95])[None, :] * (cdf[:, -1:] - 1)).astype(np.int64)
        bb = np.stack([np.searchsorted(cdf[k], want[k], side="right") for k in range(K)])
        marked = np.zeros((K, nb), dtype=bool)
        marked[np.arange(K)[:, None], bb] = True
        words = nb // 32
        mask = np.packbits(marked.reshape(K, words, 32), axis=2, bitorder="little").view(np.uint32).reshape(K, words)
        cap = int((hd * marked).sum(1).max()) + 8
        out = {}
        for name, L, rows_t, rg_t, dev in (("hip", lib, rows_d, rg_d, "cuda"), ("numpy", cpu, rows_h, rg_h, "cpu")):
            m_t = torch.from_numpy(mask.view(np.int32).copy()).to(dev)
            cand = torch.zeros((K, cap), dtype=rows_t.dtype, device=dev)
            cn = torch.zeros(K, dtype=torch.int64, device=dev)
            assert getattr(L, f"fiveeq_select_bins_{sfx}")(K, n, n, p(rows_t), p(rg_t), nb, p(m_t), p(cand), cap, p(cn), None) == 0
            below = np.concatenate([np.zeros((K, 1), dtype=np.int64), cdf[:, :-1]], axis=1)
            cb = np.concatenate([np.zeros((K, 1), dtype=np.int64), np.cumsum(hd * marked, axis=1)[:, :-1]], axis=1)
            ranks = torch.from_numpy(np.ascontiguousarray(np.take_along_axis(cb, bb, 1) + want - np.take_along_axis(below, bb, 1))).to(dev)
            picked = torch.zeros((K, 3), dtype=torch.float64, device=dev)
            assert getattr(L, f"fiveeq_select_pick_{sfx}")(K, 1, cap, p(cand), p(cn), 3, p(ranks), p(picked), None) == 0
            if dev == "cuda":
                torch.cuda.synchronize()
            out[name] = (cn.cpu().numpy(), [np.sort(cand[k, :int(cn[k])].cpu().numpy()) for k in range(K)], picked.cpu().numpy())
        if sfx == "f64":
            assert np.array_equal(out["hip"][0], out["numpy"][0]) and np.array_equal(out["hip"][2], out["numpy"][2])
            assert all(np.array_equal(a_, b_) for a_, b_ in zip(out["hip"][1], out["numpy"][1]))
        assert np.array_equal(out["hip"][0], (hd * marked).sum(1))                 # the kernels agree with their own histogram
        x64 = x.astype(np.float64)
        for k in range(K):                                                          # ... and the picks are np.sort's order statistics
            srt = np.sort(x64[k][~np.isnan(x64[k])])
            assert np.array_equal(out["hip"][2][k], srt[want[k]])


def test_a_range_too_narrow_for_fp32_constants_still_bins_every_member():
    """hi - lo = 1e-40: n_bins / (hi - lo) overflows a float.  The fp32 bin rule keeps its two constants finite, so every member
    still lands inside the histogram (below lo -> first bin, above -> last), for the scalar entry point and the ranged one."""
    from fiveeqscm_amd import _capi
    lib = _capi.load()
    n, nb = 100_000, 4096
    x = torch.linspace(-1.0, 1.0, n, device="cuda", dtype=torch.float32).reshape(1, n).contiguous()
    p = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
    h = torch.zeros((1, nb), dtype=torch.int64, device="cuda")
    _capi.check(lib, lib.fiveeq_hist_rows_f32(1, n, n, p(x), 0.0, 1e-40, nb, p(h), None))
    rg = torch.tensor([[0.0, 1e-40]], dtype=torch.float64, device="cuda")
    h2 = torch.zeros((1, nb), dtype=torch.int64, device="cuda")
    _capi.check(lib, lib.fiveeq_hist_rows_ranged_f32(1, n, n, p(x), p(rg), nb, p(h2), None))
    torch.cuda.synchronize()
    for hh in (h, h2):
        c = hh.cpu().numpy()[0]
        assert c.sum() == n and c[0] >= (x <= 0).sum().item() - 1 and c[0] + c[-1] == n


def test_summary_fuzz_on_the_device_against_numpy_percentile():
    """200 random shapes through the HIP passes: 1-5 rows of 1 to ~300k members (every load shape: below one wave, one
    16-byte stride, ragged tails), six distributions, fp32 / fp64, 1-8 random percentiles with repeats, infinite members thrown
    in — np.percentile bit for bit (NaN where NumPy's interpolation gives NaN), moments to 1e-12."""
    from fiveeqscm_amd.distributed import gather_summary
    rng = np.random.default_rng(1)
    grid = [0, 0.1, 1, 5, 25, 33.3, 50, 50.01, 75, 95, 99.9, 100]
    sizes = [1, 2, 3, 7, 63, 64, 65, 255, 256, 257, 1000, 2047, 2048, 2049, 4100, 65_537, 300_001]
    for it in range(200):
        K, n, kind = int(rng.integers(1, 6)), int(rng.choice(sizes)), int(rng.integers(0, 6))
        x = (lambda: rng.normal(size=(K, n)), lambda: rng.uniform(size=(K, n)) ** 5, lambda: np.round(rng.normal(size=(K, n)) * 2) / 2,
             lambda: rng.standard_cauchy(size=(K, n)) * 1e6, lambda: np.full((K, n), 3.25),
             lambda: rng.normal(size=(K, n)) * 1e-30 + 1e-3)[kind]()
        if rng.uniform() < 0.2 and n > 2:
            x[rng.integers(0, K), rng.integers(0, n)] = np.inf * rng.choice([-1, 1])
        xs = x.astype(np.float32 if rng.uniform() < 0.5 else np.float64)
        pct = tuple(float(v) for v in rng.choice(grid, size=int(rng.integers(1, 9))))
        out = gather_summary(torch.from_numpy(xs).cuda(), pct)
        x64 = xs.astype(np.float64)
        with np.errstate(invalid="ignore"):
            want = np.percentile(x64, pct, axis=1).T
            assert np.array_equal(out["percentiles"].numpy(), want, equal_nan=True), (it, K, n, kind, xs.dtype, pct)
            fin = np.isfinite(x64).all(axis=1)
            np.testing.assert_allclose(out["mean"].numpy()[fin], x64.mean(1)[fin], rtol=1e-12, atol=1e-12 * np.abs(x64[fin]).max(initial=1.0))
        assert np.array_equal(out["min"].numpy(), x64.min(1)) and np.array_equal(out["max"].numpy(), x64.max(1))
