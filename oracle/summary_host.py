"""The end-of-run summary (moments + exact percentiles of rows over all ranks) restated in TORCH OPS on host tensors.

TEST INFRASTRUCTURE ONLY (like everything under oracle/).  The product — fiveeqscm_amd.distributed.gather_summary /
exact_percentiles — summarises rows that live on a GPU through four HIP passes and has no CPU path; this module is the
independent second route to the same numbers that the CPU tests check the exchange logic with (gloo, world 2 and 8) and that the
GPU tests compare the passes against: a different selection algorithm (candidate VALUE intervals with a one-bin margin, exact
counts below them by comparison, a sort on the root) on a different back end.  Until round 4 it was the body of the product's
functions for host rows; only tests ever called it there.

The reference (stujen/fiveEqSCM @ v0) has no counterpart: no summary statistics, no percentiles.
"""
import torch

from fiveeqscm_amd.distributed import (SELECT_BINS, _comm_tensor, _dist, lerp, linear_positions, merge_moments,
                                       percentiles_sorted)


def local_moments(x):
    """x [K, n] -> [K, 5] = (count, mean, M2, min, max) per row, accumulated in fp64 in single passes over the rows
    without an fp64 copy of them: sum(x) and sum(x^2) (both with fp64 accumulation of the exactly converted elements),
    M2 = sum(x^2) - n mean^2.  The subtraction costs log10(mean^2 / var) digits of the variance (one digit for an
    ensemble temperature row; the kernels' own per-wave records use the same two sums)."""
    n = x.shape[1]
    s1 = _by_rows(x, lambda v, d: v.sum(dim=d, dtype=torch.float64), lambda v, d: v.sum(dim=d))
    s2 = _by_rows(x, lambda v, d: torch.linalg.vector_norm(v, ord=2, dim=d, dtype=torch.float64) ** 2, lambda v, d: v.sum(dim=d))
    mean = s1 / n
    m2 = (s2 - n * mean * mean).clamp_min(0.0)
    cnt = torch.full_like(mean, float(n))
    mn = _by_rows(x, lambda v, d: v.amin(dim=d), lambda v, d: v.amin(dim=d)).to(torch.float64)
    mx = _by_rows(x, lambda v, d: v.amax(dim=d), lambda v, d: v.amax(dim=d)).to(torch.float64)
    return torch.stack([cnt, mean, m2, mn, mx], dim=1)


_ROW_PIECE = 8192


def _by_rows(x, first, second):
    """Reduce x [K, n] along its rows in two stages — pieces of 8192 members, then the pieces — because torch reduces a long
    axis into a HANDFUL of outputs an order of magnitude slower than into many (3 rows of 12.5M members: 14 ms for the four
    moments in one stage, 1 ms in two).  `first(v, dim)` reduces the members of a piece, `second(v, dim)` combines pieces."""
    K, n = x.shape
    if n <= 4 * _ROW_PIECE:
        return first(x, 1)
    m = (n // _ROW_PIECE) * _ROW_PIECE
    parts = [first(x[:, :m].reshape(K, n // _ROW_PIECE, _ROW_PIECE), 2)]
    if n > m:
        parts.append(first(x[:, m:], 1).reshape(K, 1))
    return second(torch.cat(parts, dim=1), 1)


def _row_histograms(rows, lo, hi, n_bins):
    """counts [K, n_bins] int64 of rows [K, n] between per-row lo/hi (lists of floats), in torch ops.  The caller never relies
    on the exact bin of a value — candidates are re-selected by VALUE with a one-bin margin."""
    K, n = rows.shape
    counts = torch.zeros((K, n_bins), dtype=torch.int64, device=rows.device)
    for k in range(K):
        if hi[k] > lo[k]:
            idx = ((rows[k].to(torch.float64) - lo[k]) * (n_bins / (hi[k] - lo[k]))).floor_().clamp_(0, n_bins - 1)
            counts[k] = torch.bincount(idx.to(torch.int64), minlength=n_bins)
        else:
            counts[k, 0] = n
    return counts


def _round_up_to(dtype, t):
    """The smallest value of `dtype` that is >= t (fp64 tensor; infinities stay): comparing a row of that dtype against it
    gives what comparing against t in exact arithmetic would."""
    if dtype == torch.float64:
        return t
    r = t.to(dtype)
    return torch.where(r.to(torch.float64) < t, torch.nextafter(r, torch.full_like(r, float("inf"))), r)


SELECT_CHUNK_ELEMS = 1 << 25     # elements of `rows` compared at a time in exact_percentiles (x P boolean temporaries)


def exact_percentiles(rows, percentiles, gmin, gmax, n_total, dst=0, group=None, n_bins=SELECT_BINS, stats=None):
    """Exact percentiles (NumPy 'linear' definition) of rows [K, n_local] over all ranks by histogram selection:
    (1) per-row histograms between the global extrema, all-reduced; (2) the bins holding the wanted order statistics
    are read off the cumulative counts; (3) every rank counts its values BELOW the value interval of those bins (one
    bin of margin each side) exactly, by comparison — all-reduced — and sends the values INSIDE any of a row's
    intervals to the root (once, however many percentiles share them); (4) the root sorts the few candidates by
    (row, value) and reads every order statistic off at (index - below) past the interval's first candidate.
    Steps (3)-(4) are tensor operations over all K x P (row, percentile) pairs at once — blocks of rows of at most
    SELECT_CHUNK_ELEMS values, one host synchronisation — so all-timestep exact percentiles cost K/block iterations,
    not K x P.
    gmin/gmax [K] fp64: global extrema (from the merged moments); n_total: members over all ranks.
    Returns [K, P] fp64 on rank `dst`, None elsewhere.  `stats`, if a dict, receives bytes_to_root / allreduce_bytes."""
    dist, rank, world, exchange = _dist(group)
    rows = rows.contiguous()
    K, n_local = rows.shape
    P = len(percentiles)
    dev = rows.device
    f64 = torch.float64
    lo_t = gmin.to(device=dev, dtype=f64).reshape(K)
    hi_t = gmax.to(device=dev, dtype=f64).reshape(K)
    counts = _row_histograms(rows, lo_t.tolist(), hi_t.tolist(), n_bins)
    if exchange:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    cdf = torch.cumsum(counts, dim=1)                                              # [K, n_bins], last = n_total
    i0_np, i1_np, gamma_np = linear_positions(n_total, percentiles)               # order-statistic indices, NumPy's rule
    i0, i1 = i0_np.tolist(), i1_np.tolist()
    frac = torch.from_numpy(gamma_np).to(device=dev, dtype=f64)
    i0_t = torch.tensor(i0, dtype=torch.int64, device=dev)
    i1_t = torch.tensor(i1, dtype=torch.int64, device=dev)
    want = torch.stack([i0_t, i1_t], dim=1).reshape(1, 2 * P)
    b = torch.searchsorted(cdf, want.expand(K, 2 * P).contiguous(), right=True).clamp_(max=n_bins - 1)
    b0, b1 = b[:, 0::2], b[:, 1::2]                                                # [K, P]: first bin with cdf > index
    # value interval of the candidate bins, widened by one bin on each side (the histogram back ends may put a value
    # that sits on a bin edge on either side of it); open-ended at the extremes; EMPTY (+inf, +inf) for a constant row,
    # whose answer is that constant
    live = (hi_t > lo_t).reshape(K, 1)
    w = ((hi_t - lo_t) / n_bins).reshape(K, 1)
    inf = torch.full((), float("inf"), dtype=f64, device=dev)
    v_lo = torch.where((b0 - 1 > 0) & (w > 0), lo_t.reshape(K, 1) + (b0 - 1).to(f64) * w, -inf)
    v_hi = torch.where((b1 + 2 < n_bins) & (w > 0), lo_t.reshape(K, 1) + (b1 + 2).to(f64) * w, inf)
    v_lo, v_hi = torch.where(live, v_lo, inf), torch.where(live, v_hi, inf)
    # The rows are compared in their OWN dtype (no fp64 copy of a 12.5M-member row): for a value x of that dtype and an fp64
    # threshold t, x < t  <=>  x < up(t) and x >= t  <=>  x >= up(t), up(t) = the smallest value of the dtype that is >= t.
    # One binary search per value against the row's 2P sorted thresholds gives i = #thresholds <= x; then
    #   x < e  <=>  i <= (#thresholds < e),      so the counts below every threshold come from ONE bincount of i,
    # and "inside any interval" is a lookup of i in a 2P+1-entry table per row.  (Boolean reductions over [rows, P, members]
    # were tried first: torch reduces a bool tensor along an axis ten times slower than it histograms an index.)
    t_lo, t_hi = _round_up_to(rows.dtype, v_lo), _round_up_to(rows.dtype, v_hi)
    E2 = 2 * P
    edges = torch.cat([t_lo, t_hi], dim=1).contiguous()                            # [K, 2P]
    es = torch.sort(edges, dim=1).values.contiguous()
    n_less = torch.searchsorted(es, edges, right=False)                            # [K, 2P]: thresholds strictly below each one
    nl_lo, nl_hi = n_less[:, :P], n_less[:, P:]
    slot = torch.arange(E2 + 1, device=dev).reshape(1, E2 + 1, 1)
    table = ((slot > nl_lo.unsqueeze(1)) & (slot <= nl_hi.unsqueeze(1))).any(dim=2)     # [K, 2P+1]: slot i lies in some interval
    below = torch.zeros((K, P), dtype=torch.int64, device=dev)
    parts, row_sizes = [], torch.zeros(K, dtype=torch.int64, device=dev)
    kb = max(1, min(K, SELECT_CHUNK_ELEMS // max(n_local, 1)))
    for k0 in range(0, K, kb):
        k1 = min(K, k0 + kb)
        x = rows[k0:k1]
        i = torch.searchsorted(es[k0:k1], x, right=True)                           # [kb, n] in 0..2P; a NaN sorts last (2P): in no
        i += (torch.arange(k1 - k0, device=dev) * (E2 + 1)).unsqueeze(1)           # interval and below nothing
        c = torch.bincount(i.flatten(), minlength=(k1 - k0) * (E2 + 1)).reshape(k1 - k0, E2 + 1)
        below[k0:k1] = torch.cumsum(c, dim=1).gather(1, nl_lo[k0:k1])              # members with i <= #thresholds < lo_j
        inside = table[k0:k1].flatten()[i]                                          # [kb, n]
        row_sizes[k0:k1] = torch.bincount(i.flatten()[inside.flatten()] // (E2 + 1), minlength=k1 - k0)
        parts.append(x[inside])                                                     # row-major: row k's candidates are contiguous
    payload = torch.cat(parts) if parts else rows.new_empty(0)
    if exchange:
        dist.all_reduce(below, op=dist.ReduceOp.SUM, group=group)
        all_sizes = [torch.empty_like(row_sizes) for _ in range(world)]
        dist.all_gather(all_sizes, row_sizes, group=group)
        all_sizes = torch.stack(all_sizes)                                         # [world, K]
        per_rank = all_sizes.sum(dim=1).cpu()
        longest = int(per_rank.max().item())
        send = torch.cat([payload, payload.new_zeros(longest - payload.numel())])
        recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, recv, dst=dst, group=group)
        if stats is not None:
            stats["bytes_to_root"] = int((per_rank.sum() - per_rank[dst]).item()) * payload.element_size()
            stats["allreduce_bytes"] = counts.numel() * 8 + below.numel() * 8     # what every rank contributes
        if rank != dst:
            return None
        ar = torch.arange(K, device=dev)
        cand = torch.cat([recv[w_][:int(per_rank[w_])] for w_ in range(world)])
        cand_row = torch.cat([torch.repeat_interleave(ar, all_sizes[w_]) for w_ in range(world)])
        seg = all_sizes.sum(dim=0)                                                 # candidates per row over all ranks
    else:
        if stats is not None:
            stats["bytes_to_root"] = 0
            stats["allreduce_bytes"] = 0
        cand, seg = payload, row_sizes
        cand_row = torch.repeat_interleave(torch.arange(K, device=dev), row_sizes)
    # sort by (row, value): by value, then stably by row
    cand = cand.to(f64)
    order = torch.argsort(cand)
    cand, cand_row = cand[order], cand_row[order]
    order = torch.argsort(cand_row, stable=True)
    cand, cand_row = cand[order], cand_row[order]
    seg_end = torch.cumsum(seg, dim=0)
    seg_start = seg_end - seg
    # candidates of the row that lie below each interval (they belong to another percentile's interval)
    skipped = torch.zeros((K, P), dtype=torch.int64, device=dev)
    if cand.numel():
        skipped.index_add_(0, cand_row, (cand.unsqueeze(1) < v_lo[cand_row]).to(torch.int64))
    at0 = seg_start.reshape(K, 1) + skipped + (i0_t.reshape(1, P) - below)
    at1 = seg_start.reshape(K, 1) + skipped + (i1_t.reshape(1, P) - below)
    ok = (at0 >= seg_start.reshape(K, 1)) & (at0 <= at1) & (at1 < seg_end.reshape(K, 1))
    c0 = cand[at0.clamp(0, max(cand.numel() - 1, 0))] if cand.numel() else torch.zeros((K, P), dtype=f64, device=dev)
    c1 = cand[at1.clamp(0, max(cand.numel() - 1, 0))] if cand.numel() else torch.zeros((K, P), dtype=f64, device=dev)
    ok &= (c0 >= v_lo) & (c1 < v_hi)                                               # both inside the interval they were sought in
    bad = live & ~ok
    if bool(bad.any().item()):
        k, j = [int(v) for v in torch.nonzero(bad)[0].tolist()]
        raise RuntimeError(f"percentile selection lost its order statistic (row {k}, p={percentiles[j]}): "
                           f"{int(at0[k, j] - seg_start[k])},{int(at1[k, j] - seg_start[k])} of {int(seg[k])} candidates")
    out = lerp(c0, c1, frac.reshape(1, P))
    return torch.where(live, out, lo_t.reshape(K, 1).expand(K, P))


def gather_summary(rows, percentiles=(5.0, 50.0, 95.0), dst=0, group=None, stats=None):
    """rows [K, n_local] on the HOST: this rank's members at K output times.  Collective over `group` (gloo).  The dict of
    fiveeqscm_amd.distributed.gather_summary: merged moments on every rank, 'percentiles' [K, P] on rank `dst` (None
    elsewhere)."""
    dist, rank, world, exchange = _dist(group)
    rows = _comm_tensor(dist, group, rows.contiguous())
    mom = local_moments(rows)
    if exchange:
        parts = [torch.empty_like(mom) for _ in range(world)]
        dist.all_gather(parts, mom, group=group)
        mom = merge_moments(torch.stack(parts))
    n_total = int(round(float(mom[0, 0].item())))
    if not exchange and rows.shape[1] <= (1 << 21):
        # nothing to exchange and a moderate row: a sort is as fast as anything
        if stats is not None:
            stats["bytes_to_root"], stats["allreduce_bytes"] = 0, 0
        pct = percentiles_sorted(torch.sort(rows, dim=1).values.to(torch.float64), percentiles)
    else:
        pct = exact_percentiles(rows, percentiles, mom[:, 3], mom[:, 4], n_total, dst=dst, group=group, stats=stats)
    return {"count": mom[:, 0], "mean": mom[:, 1], "var": mom[:, 2] / mom[:, 0], "min": mom[:, 3], "max": mom[:, 4],
            "percentiles": pct}
