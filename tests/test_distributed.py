"""The N>1 path on CPU: world_size-2 (and 3, 8) gloo processes exercise sharding + the end-of-run summary exchange — the
product's host logic (fiveeqscm_amd.distributed: which bins are marked, the rank bookkeeping, the exchanges) with the four HIP
passes replaced by their NumPy restatement (oracle/summary_passes.py), i.e. the same code that runs over RCCL on the GPUs; and,
as a second route to the same numbers, the torch-ops restatement of the whole summary (oracle/summary_host.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from fiveeqscm_amd.distributed import (gather_summary, histogram_percentiles, merge_moments, moments_from_sums,
                                       percentiles_sorted, reduce_stats, shard_bounds)
from oracle import summary_host
from oracle.summary_host import local_moments


def _use_oracle_passes():
    """The product's summary takes device rows through the HIP passes; here (no GPU) the passes are the NumPy restatement
    behind the same signatures, and host rows are let in."""
    import ctypes

    from fiveeqscm_amd import distributed
    from oracle.summary_passes import SummaryPasses

    class _Check:
        @staticmethod
        def check(lib, rc):
            assert rc == 0

    passes = SummaryPasses()
    distributed._lib_and_stream = lambda rows: (passes, _Check, ctypes, None)
    distributed._passes_apply = lambda rows: rows.dtype in (torch.float32, torch.float64)


@pytest.fixture
def numpy_passes():
    from fiveeqscm_amd import distributed
    saved = distributed._lib_and_stream, distributed._passes_apply
    _use_oracle_passes()
    yield
    distributed._lib_and_stream, distributed._passes_apply = saved


def test_host_rows_are_refused_by_the_product():
    """No CPU fallback: rows on the host raise (the torch-ops restatement lives under oracle/, for tests)."""
    x = torch.arange(12, dtype=torch.float64).reshape(2, 6)
    with pytest.raises(TypeError, match="no CPU fallback"):
        gather_summary(x, percentiles=(50.0,))
    from fiveeqscm_amd.distributed import exact_percentiles
    with pytest.raises(TypeError, match="no CPU fallback"):
        exact_percentiles(x, (50.0,), x.min(1).values, x.max(1).values, 6)


def test_shard_bounds_cover_and_balance():
    for n, w in ((10, 3), (1_000_000, 8), (7, 8), (10_000_000, 8), (5, 1)):
        b = [shard_bounds(n, r, w) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 3, 3)


def test_moments_merge_equals_global():
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.normal(2.0, 3.0, size=(4, 1001)))
    parts = torch.stack([local_moments(x[:, :300]), local_moments(x[:, 300:777]), local_moments(x[:, 777:])])
    m = merge_moments(parts)
    np.testing.assert_allclose(m[:, 1].numpy(), x.numpy().mean(1), rtol=1e-13)
    np.testing.assert_allclose((m[:, 2] / m[:, 0]).numpy(), x.numpy().var(1), rtol=1e-12)
    assert torch.equal(m[:, 3], x.min(1).values) and torch.equal(m[:, 4], x.max(1).values)


def test_percentiles_match_numpy():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(3, 997))
    xs = torch.sort(torch.from_numpy(x), dim=1).values
    got = percentiles_sorted(xs, (0, 5, 50, 95, 100, 33.3)).numpy()
    np.testing.assert_allclose(got, np.percentile(x, (0, 5, 50, 95, 100, 33.3), axis=1).T, rtol=1e-13)


def test_moments_from_sums_single_process():
    x = torch.tensor([[1.0, 2.0, 3.0, 6.0]], dtype=torch.float64)
    sums = torch.tensor([[4.0, 12.0, 50.0, 1.0, 6.0]], dtype=torch.float64)
    m = reduce_stats(sums)            # no process group: local only
    assert m["mean"].item() == 3.0 and abs(m["var"].item() - x.var(unbiased=False).item()) < 1e-15
    assert moments_from_sums(sums)["max"].item() == 6.0


def _np_hist(x, lo, hi, n_bins):
    pos = np.floor((x - lo) * (n_bins / (hi - lo))).astype(np.int64).clip(0, n_bins - 1)
    return np.stack([np.bincount(r, minlength=n_bins) for r in pos])


def test_histogram_percentiles_within_one_bin_width():
    rng = np.random.default_rng(3)
    x = rng.normal(2.0, 0.8, size=(5, 200_000))
    lo, hi, nb = -3.0, 8.0, 4096
    h = torch.from_numpy(_np_hist(x, lo, hi, nb))
    got, total = histogram_percentiles(h, lo, hi, (1.0, 5.0, 50.0, 95.0, 99.0))
    want = np.percentile(x, (1.0, 5.0, 50.0, 95.0, 99.0), axis=1).T
    assert total.tolist() == [200_000.0] * 5
    assert np.abs(got.numpy() - want).max() < (hi - lo) / nb          # below one bin width (2.7 mK here)
    assert np.abs(got.numpy() - want)[:, 1:4].max() < 4e-4            # in-bin interpolation: better where bins are full


def test_single_process_summary_needs_no_process_group(numpy_passes):
    x = torch.arange(12, dtype=torch.float64).reshape(2, 6)
    for fn in (gather_summary, summary_host.gather_summary):
        s = fn(x, percentiles=(50.0,))
        assert s["percentiles"][:, 0].tolist() == [2.5, 8.5] and s["count"].tolist() == [6.0, 6.0]


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    _use_oracle_passes()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(42)
        full = rng.normal(1.5, 0.7, size=(3, n_total))            # every rank can rebuild the global T rows
        lo, hi = shard_bounds(n_total, rank, world)
        s = gather_summary(torch.from_numpy(full[:, lo:hi].copy()), percentiles=(5.0, 50.0, 95.0))
        close = (49.9, 50.0, 50.0, 50.1, 0.0, 100.0)              # overlapping candidate intervals, a duplicate, the extremes
        s2 = gather_summary(torch.from_numpy(full[:, lo:hi].copy()), percentiles=close)
        h2 = summary_host.gather_summary(torch.from_numpy(full[:, lo:hi].copy()), percentiles=close)   # the second route
        x = torch.from_numpy(full[:, lo:hi].copy())
        sums = torch.stack([torch.full((3,), float(hi - lo), dtype=torch.float64), x.sum(1), (x * x).sum(1),
                            x.min(1).values, x.max(1).values], dim=1)
        m = reduce_stats(sums)                                     # every rank gets the ensemble moments
        stats_ok = (np.allclose(m["mean"].numpy(), full.mean(1), rtol=1e-13)
                    and np.allclose(m["var"].numpy(), full.var(1), rtol=1e-10)
                    and np.array_equal(m["min"].numpy(), full.min(1)) and np.array_equal(m["max"].numpy(), full.max(1))
                    and m["count"].tolist() == [float(n_total)] * 3)
        hloc = torch.from_numpy(_np_hist(full[:, lo:hi], -3.0, 6.0, 2048))
        hp, htot = histogram_percentiles(hloc, -3.0, 6.0, (5.0, 50.0, 95.0))          # all-reduce(SUM) of the counts
        stats_ok = bool(stats_ok and htot.tolist() == [float(n_total)] * 3
                        and np.abs(hp.numpy() - np.percentile(full, (5.0, 50.0, 95.0), axis=1).T).max() < 9.0 / 2048)
        if rank == 0:
            want = np.percentile(full, (5.0, 50.0, 95.0), axis=1).T
            ok = (np.allclose(s["percentiles"].numpy(), want, rtol=1e-13)
                  and np.allclose(s["mean"].numpy(), full.mean(1), rtol=1e-13)
                  and np.allclose(s["var"].numpy(), full.var(1), rtol=1e-12)
                  and s["count"].tolist() == [float(n_total)] * 3
                  and np.array_equal(s["min"].numpy(), full.min(1)) and np.array_equal(s["max"].numpy(), full.max(1)))
            ok = ok and np.array_equal(s2["percentiles"].numpy(), np.percentile(full, close, axis=1).T)      # bit for bit
            ok = ok and np.allclose(h2["percentiles"].numpy(), s2["percentiles"].numpy(), rtol=1e-13)
            q.put(bool(ok and stats_ok))
        else:
            q.put(bool(stats_ok and s["percentiles"] is None and s2["percentiles"] is None and h2["percentiles"] is None
                       and abs(float(s["mean"][0]) - full[0].mean()) < 1e-12))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [1000, 1001])        # even split and ragged split (padding path)
def test_gloo_world2_summary_exchange(n_total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(results)


def test_selection_percentiles_edge_cases():
    """The torch-ops restatement (oracle/summary_host.py): exact_percentiles (its multi-rank path, called directly) and
    gather_summary (one rank: sort): ties, constant rows, tiny rows, fp32 rows kept in fp32 on the wire."""
    from oracle.summary_host import exact_percentiles, gather_summary
    rng = np.random.default_rng(5)
    for n in (1, 2, 5, 1000, 200_001):
        x = np.stack([rng.normal(size=n), rng.uniform(size=n) ** 3, np.full(n, 2.5), np.round(rng.normal(size=n), 1)])
        for dt in (np.float64, np.float32):
            xs = x.astype(dt)
            st = {}
            out = gather_summary(torch.from_numpy(xs), (0.0, 5.0, 50.0, 95.0, 100.0, 33.3), stats=st)
            want = np.percentile(xs.astype(np.float64), (0.0, 5.0, 50.0, 95.0, 100.0, 33.3), axis=1).T
            np.testing.assert_allclose(out["percentiles"].numpy(), want, rtol=1e-14, atol=0)
            assert st["bytes_to_root"] == 0
            t = torch.from_numpy(xs)
            sel = exact_percentiles(t, (0.0, 5.0, 50.0, 95.0, 100.0, 33.3), t.min(1).values.double(), t.max(1).values.double(), n)
            np.testing.assert_allclose(sel.numpy(), want, rtol=1e-14, atol=0)


def test_selection_percentiles_many_rows_in_blocks(monkeypatch):
    """All-timestep shape of the torch-ops exact_percentiles: many rows, processed in blocks of rows (the block size forced
    small so that a block boundary falls inside the loop), percentiles so close that their candidate intervals overlap (a
    row's candidates travel once), a constant row in the middle."""
    from oracle import summary_host as distributed
    from oracle.summary_host import exact_percentiles
    rng = np.random.default_rng(11)
    K, n = 37, 30_011
    x = rng.normal(size=(K, n)) * rng.uniform(0.1, 5.0, size=(K, 1)) + rng.normal(size=(K, 1))
    x[17] = -3.25
    pct = (0.0, 5.0, 49.999, 50.0, 50.001, 95.0, 100.0)
    want = np.percentile(x, pct, axis=1).T
    t = torch.from_numpy(x)
    for chunk in (1 << 25, 5 * n + 1, 1):
        monkeypatch.setattr(distributed, "SELECT_CHUNK_ELEMS", chunk)
        st = {}
        got = exact_percentiles(t, pct, t.min(1).values, t.max(1).values, n, stats=st)
        np.testing.assert_allclose(got.numpy(), want, rtol=1e-14, atol=0)
    x32 = x.astype(np.float32)
    t32 = torch.from_numpy(x32)
    got = exact_percentiles(t32, pct, t32.min(1).values.double(), t32.max(1).values.double(), n)
    np.testing.assert_allclose(got.numpy(), np.percentile(x32.astype(np.float64), pct, axis=1).T, rtol=1e-14, atol=0)


# ---- BASELINE configs[3] rehearsed at world size 8 (CPU, gloo): every rank computes ONLY its shard of the
# shard-computable Latin hypercube, advances it with the C oracle (no GPU here), and the summary exchange must
# reproduce np.percentile / mean of the ORACLE's T over the whole ensemble computed in one piece. -------------
N8, STEPS8, YEARS8 = 40_003, 110, [30, 70, 109]


def _worker8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    import torch.distributed as dist
    from fiveeqscm_amd import emissions, params
    from oracle import c_oracle
    _use_oracle_passes()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        base = params.default_params("multigas")
        E = emissions.rcp_like_emissions(750, 3)[200:200 + STEPS8]
        lo, hi = shard_bounds(N8, rank, world)
        mine = params.sample_ensemble_shard(base, N8, lo, hi)                 # O(shard): nothing of size N8 here
        T = c_oracle.run(E, mine, hi - lo, keep=("T",))["T"]
        st = {}
        s = gather_summary(torch.from_numpy(np.ascontiguousarray(T[YEARS8])), percentiles=(5.0, 50.0, 95.0), stats=st)
        if rank == 0:
            whole = params.sample_ensemble_shard(base, N8)
            Tw = c_oracle.run(E, whole, N8, keep=("T",), n_threads=4)["T"][YEARS8]
            ok = (np.allclose(s["percentiles"].numpy(), np.percentile(Tw, (5.0, 50.0, 95.0), axis=1).T, rtol=1e-13)
                  and np.allclose(s["mean"].numpy(), Tw.mean(1), rtol=1e-13)
                  and np.allclose(s["var"].numpy(), Tw.var(1), rtol=1e-11)
                  and np.array_equal(s["min"].numpy(), Tw.min(1)) and np.array_equal(s["max"].numpy(), Tw.max(1))
                  and s["count"].tolist() == [float(N8)] * 3
                  and st["bytes_to_root"] < 0.05 * 3 * N8 * 8)                # a few % of gathering the rows
            q.put(bool(ok))
        else:
            q.put(s["percentiles"] is None)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_gloo_world8_config4_rehearsal_against_the_oracle():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(results)


def test_forced_collectives_in_a_one_rank_group(numpy_passes):
    """force_collectives: in a process group of ONE rank every collective of the summary exchange executes on the
    backend instead of being skipped — here over gloo; tests/test_distributed_gpu.py runs the same over RCCL on one
    GPU.  Results must equal the no-process-group answers."""
    import torch.distributed as dist

    from fiveeqscm_amd import distributed as D
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rng = np.random.default_rng(5)
    rows = torch.from_numpy(rng.normal(1.5, 0.7, size=(3, 40_001)))
    sums = torch.stack([torch.full((3,), 40_001.0, dtype=torch.float64), rows.sum(1), (rows * rows).sum(1),
                        rows.min(1).values, rows.max(1).values], dim=1)
    hist = torch.from_numpy(_np_hist(rows.numpy(), -2.0, 5.0, 512))
    want = gather_summary(rows, percentiles=(5.0, 50.0, 95.0))
    want_m, want_h = reduce_stats(sums), histogram_percentiles(hist, -2.0, 5.0)
    calls = []
    real = {name: getattr(dist, name) for name in ("all_reduce", "all_gather", "gather")}

    def spy(name):
        def f(*a, **k):
            calls.append(name)
            return real[name](*a, **k)
        return f

    dist.init_process_group("gloo", rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{port}")
    prev = D.force_collectives(True)
    try:
        for name in real:
            setattr(dist, name, spy(name))
        st = {}
        got = gather_summary(rows, percentiles=(5.0, 50.0, 95.0), stats=st)
        got_m, got_h = reduce_stats(sums), histogram_percentiles(hist, -2.0, 5.0)
    finally:
        for name, fn in real.items():
            setattr(dist, name, fn)
        D.force_collectives(prev)
        dist.destroy_process_group()
    # summary: moments all_gather, histogram all_reduce, candidate counts all_gather, candidates gather; then reduce_stats'
    # 3 all_reduces and histogram_percentiles' one
    assert calls.count("all_gather") == 2 and calls.count("gather") == 1 and calls.count("all_reduce") == 1 + 3 + 1
    np.testing.assert_allclose(got["percentiles"].numpy(), np.percentile(rows.numpy(), (5.0, 50.0, 95.0), axis=1).T,
                               rtol=1e-13)
    np.testing.assert_allclose(got["percentiles"].numpy(), want["percentiles"].numpy(), rtol=1e-13)
    assert st["bytes_to_root"] == 0 and st["allreduce_bytes"] > 0
    for k in ("mean", "var", "min", "max", "count"):
        assert torch.equal(got_m[k], want_m[k])
        np.testing.assert_allclose(got[k].numpy(), want[k].numpy(), rtol=1e-12)
    assert torch.equal(got_h[0], want_h[0]) and torch.equal(got_h[1], want_h[1])


# ---- the summary's PASS-BASED path (what device rows take: moments / ranged histogram / bin-mask selection / rank pick, host
# bookkeeping in between) run on CPU tensors against the NumPy restatement of the four passes (oracle/summary_passes.py): the
# container without a GPU still exercises which bins are marked, the rank arithmetic, the packed upload, and the multi-rank
# exchange with one candidate segment per rank on the root -------------------------------------------------------------------
def _worker_passes(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    _use_oracle_passes()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(4242)
        full = np.stack([rng.normal(1.5, 0.7, size=n_total), rng.uniform(size=n_total) ** 3, np.full(n_total, 2.5),
                         np.round(rng.normal(size=n_total), 1), rng.standard_cauchy(size=n_total)])
        lo, hi = shard_bounds(n_total, rank, world)
        pct = (0.0, 5.0, 49.9, 50.0, 50.0, 50.1, 95.0, 100.0)
        ok = True
        for dt in (np.float64, np.float32):
            mine = torch.from_numpy(full[:, lo:hi].astype(dt))
            st = {}
            s = gather_summary(mine, percentiles=pct, stats=st)
            if rank == 0:
                x = full.astype(dt).astype(np.float64)
                ok = ok and np.array_equal(s["percentiles"].numpy(), np.percentile(x, pct, axis=1).T)      # bit for bit
                ok = ok and np.allclose(s["mean"].numpy(), x.mean(1), rtol=1e-12) and np.allclose(s["var"].numpy(), x.var(1), rtol=1e-9)
                ok = ok and np.array_equal(s["min"].numpy(), x.min(1)) and np.array_equal(s["max"].numpy(), x.max(1))
                ok = ok and s["count"].tolist() == [float(n_total)] * 5 and st["allreduce_bytes"] == 5 * 4096 * 8
                ok = ok and (world == 1 or 0 < st["bytes_to_root"])
            else:
                ok = ok and s["percentiles"] is None and abs(float(s["mean"][0]) - full[0].astype(dt).astype(np.float64).mean()) < 1e-6
        # the heavy-ties fallback (rows summarised in halves) must still account for what travelled (the caller's
        # bytes_to_root / bytes_to_root_per_rank stayed 0 although candidates did travel)
        from fiveeqscm_amd import distributed as D
        saved, D.SELECT_CAND_BYTES = D.SELECT_CAND_BYTES, 2_000
        try:
            mine = torch.from_numpy(full[:, lo:hi])
            st, st_whole = {}, {}
            s = gather_summary(mine, percentiles=pct, stats=st)
            D.SELECT_CAND_BYTES = saved
            gather_summary(mine, percentiles=pct, stats=st_whole)
            ok = ok and st.get("split_rows") is True and "split_rows" not in st_whole
            ok = ok and len(st["bytes_to_root_per_rank"]) == world and sum(st["bytes_to_root_per_rank"]) == st["bytes_to_root"]
            ok = ok and st["bytes_to_root_per_rank"][0] == 0 and st["allreduce_bytes"] > st_whole["allreduce_bytes"]
            ok = ok and (world == 1 or (st["bytes_to_root"] > 0 and all(v > 0 for v in st["bytes_to_root_per_rank"][1:])))
            # every half selects from its OWN rows' bins: together they send exactly what the unsplit summary sends
            ok = ok and st["bytes_to_root"] == st_whole["bytes_to_root"]
            if rank == 0:
                ok = ok and np.array_equal(s["percentiles"].numpy(), np.percentile(full, pct, axis=1).T)
        finally:
            D.SELECT_CAND_BYTES = saved
        q.put(bool(ok))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 1001), (2, 4000), (8, 40_003)])
def test_pass_based_summary_over_gloo_on_cpu(world, n_total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_passes, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(results)


def test_pass_based_summary_edge_cases_on_cpu(monkeypatch):
    """One process, no group: ties, constant rows, heavy tails, a NaN row (-> NaN, like np.percentile), rows of 1, 2, 5
    members, a ragged size — the selection logic of the device path, on the NumPy restatement of its passes."""
    from fiveeqscm_amd import distributed
    from fiveeqscm_amd.distributed import exact_percentiles
    saved = distributed._lib_and_stream, distributed._passes_apply
    try:
        _use_oracle_passes()
        rng = np.random.default_rng(9)
        pct = (0.0, 5.0, 33.3, 49.999, 50.0, 50.001, 95.0, 100.0)
        for n in (1, 2, 5, 64, 1000, 70_001):
            x = np.stack([rng.normal(1.8, 0.6, size=n), rng.uniform(size=n) ** 3, np.full(n, 2.5), np.round(rng.normal(size=n), 1),
                          rng.standard_cauchy(size=n), np.where(rng.uniform(size=n) < 0.9, 0.25, rng.normal(size=n))])
            for dt in (np.float64, np.float32):
                xs = x.astype(dt)
                out = gather_summary(torch.from_numpy(xs), pct)
                x64 = xs.astype(np.float64)
                want = np.percentile(x64, pct, axis=1).T
                assert np.array_equal(out["percentiles"].numpy(), want), f"n={n} {dt}"        # np.percentile BIT FOR BIT
                np.testing.assert_allclose(out["mean"].numpy(), x64.mean(1), rtol=1e-12, atol=1e-13)
                sel = exact_percentiles(torch.from_numpy(xs), pct, torch.from_numpy(x64.min(1)), torch.from_numpy(x64.max(1)), n)
                assert np.array_equal(sel.numpy(), want)
        # a row of heavy ties beside ordinary rows: past SELECT_CAND_BYTES the rows are summarised in halves, recursively
        monkeypatch.setattr(distributed, "SELECT_CAND_BYTES", 20_000)
        z = rng.normal(size=(7, 3000))
        z[2] = np.where(rng.uniform(size=3000) < 0.95, 1.25, z[2])
        z[5, 11] = np.nan
        got = gather_summary(torch.from_numpy(z), (5.0, 50.0, 95.0))["percentiles"].numpy()
        ok_rows = [0, 1, 2, 3, 4, 6]
        np.testing.assert_allclose(got[ok_rows], np.percentile(z[ok_rows], (5.0, 50.0, 95.0), axis=1).T, rtol=1e-14, atol=0)
        assert np.isnan(got[5]).all()
        monkeypatch.setattr(distributed, "SELECT_CAND_BYTES", 4 << 30)
        y = rng.normal(size=(2, 5000))
        y[1, 77] = np.nan
        out = gather_summary(torch.from_numpy(y), (5.0, 50.0, 95.0))
        np.testing.assert_allclose(out["percentiles"].numpy()[0], np.percentile(y[0], (5.0, 50.0, 95.0)), rtol=1e-14)
        assert np.isnan(out["percentiles"].numpy()[1]).all() and np.isnan(out["mean"][1].item())
        assert out["min"][1].item() == np.nanmin(y[1]) and out["max"][1].item() == np.nanmax(y[1])
    finally:
        distributed._lib_and_stream, distributed._passes_apply = saved


def test_pass_based_summary_fuzz_against_numpy_percentile():
    """300 random shapes: 1-5 rows of 1-5000 members, six distributions (incl. constant rows, heavy ties, values of 1e-30
    spread, 1e6-scaled Cauchy), fp32 and fp64, 1-8 random percentiles with repeats, infinite members thrown in — the result is
    np.percentile's bit for bit (NaN where NumPy's interpolation gives NaN)."""
    from fiveeqscm_amd import distributed
    saved = distributed._lib_and_stream, distributed._passes_apply
    try:
        _use_oracle_passes()
        rng = np.random.default_rng(0)
        grid = [0, 0.1, 1, 5, 25, 33.3, 50, 50.01, 75, 95, 99.9, 100]
        for it in range(300):
            K, n, kind = int(rng.integers(1, 6)), int(rng.choice([1, 2, 3, 7, 64, 65, 100, 1000, 5000])), int(rng.integers(0, 6))
            x = (lambda: rng.normal(size=(K, n)), lambda: rng.uniform(size=(K, n)) ** 5, lambda: np.round(rng.normal(size=(K, n)) * 2) / 2,
                 lambda: rng.standard_cauchy(size=(K, n)) * 1e6, lambda: np.full((K, n), 3.25),
                 lambda: rng.normal(size=(K, n)) * 1e-30 + 1e-3)[kind]()
            if rng.uniform() < 0.2 and n > 2:
                x[rng.integers(0, K), rng.integers(0, n)] = np.inf * rng.choice([-1, 1])
            xs = x.astype(np.float32 if rng.uniform() < 0.5 else np.float64)
            pct = tuple(float(v) for v in rng.choice(grid, size=int(rng.integers(1, 9))))
            got = gather_summary(torch.from_numpy(xs), pct)["percentiles"].numpy()
            with np.errstate(invalid="ignore"):
                want = np.percentile(xs.astype(np.float64), pct, axis=1).T
            assert np.array_equal(got, want, equal_nan=True), (it, K, n, kind, xs.dtype, pct)
    finally:
        distributed._lib_and_stream, distributed._passes_apply = saved


def _worker_fuzz(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    _use_oracle_passes()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(77)                       # the same stream on every rank: each rebuilds the global rows
        grid = [0, 1, 5, 33.3, 50, 50.01, 95, 99.9, 100]
        bad = []
        for it in range(60):
            K, n_total, kind = int(rng.integers(1, 5)), int(rng.choice([1, 2, 3, 5, 64, 200, 3001])), int(rng.integers(0, 4))
            full = (lambda: rng.normal(size=(K, n_total)), lambda: np.round(rng.normal(size=(K, n_total))),
                    lambda: rng.standard_cauchy(size=(K, n_total)), lambda: np.full((K, n_total), -1.5))[kind]()
            dt = np.float32 if rng.uniform() < 0.5 else np.float64
            pct = tuple(float(v) for v in rng.choice(grid, size=int(rng.integers(1, 6))))
            lo, hi = shard_bounds(n_total, rank, world)       # n_total < world: some shards are EMPTY
            s = gather_summary(torch.from_numpy(np.ascontiguousarray(full[:, lo:hi].astype(dt))), percentiles=pct)
            x = full.astype(dt).astype(np.float64)
            if rank == 0:
                if not (np.array_equal(s["percentiles"].numpy(), np.percentile(x, pct, axis=1).T)
                        and np.allclose(s["mean"].numpy(), x.mean(1), rtol=1e-12, atol=1e-12 * max(1.0, np.abs(x[np.isfinite(x)]).max()))
                        and np.array_equal(s["min"].numpy(), x.min(1)) and s["count"].tolist() == [float(n_total)] * K):
                    bad.append((it, K, n_total, kind, dt.__name__, pct))
            elif s["percentiles"] is not None or s["count"].tolist() != [float(n_total)] * K:
                bad.append((it, "non-root"))
        q.put(bad)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_pass_based_summary_fuzz_over_three_ranks_incl_empty_shards():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_fuzz, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == [[], [], []], results
