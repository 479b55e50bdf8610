"""Multi-GPU side of the engine: member sharding and the end-of-run summary exchange.

Ensemble members never interact, so the ensemble is cut into contiguous member ranges, one per
GPU / process, and time-stepping needs NO communication.  The only exchange is after the last
step: summary statistics of T (or C) at selected output times, over ALL members —
  * moments (count, mean, variance, min, max): each rank reduces its shard, the tiny per-rank
    records are all-gathered and merged with Chan's parallel-variance formula;
  * exact percentiles — one rank: a device sort of its rows; several ranks: by SELECTION, so that the ensemble does
    not have to travel: a 4096-bin histogram per output time
    between the global min and max is all-reduced (0.1 MB per rank for three output times), the bins
    holding the wanted order statistics are located on its cumulative counts, and only the members
    inside those bins travel to the root (a few thousand values out of 10M), where they are sorted
    and read off with NumPy's default linear interpolation.  Values keep their dtype on the wire.
The summary is four HIP passes over device rows (include/fiveeq.h "END-OF-RUN SUMMARY": moments, histogram with per-row
device ranges, selection with ballot counts and LDS compaction, radix pick) with the small bookkeeping between them done
on the host in NumPy — the rows are read three times at memory speed and nothing else of their size exists.  Rows on the
host are refused: there is no CPU path (a torch-ops restatement for tests: oracle/summary_host.py).
`torch.distributed` backend "nccl" is RCCL on ROCm; tests/test_distributed.py runs the same host logic over gloo with the
passes replaced by their NumPy restatement.  The reference has no distributed code at all.
"""
import os

import numpy as np
import torch


def shard_bounds(n_total, rank, world):
    """Contiguous, balanced member range [lo, hi) of `rank` (sizes differ by at most one)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside 0..{world - 1}")
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


# First-contact hook: with a process group of ONE rank the collectives below would all be skipped ("nothing to
# exchange").  force_collectives(True) — or FIVEEQ_FORCE_COLLECTIVES=1 in the environment — makes every one of them
# execute on the group's backend anyway, so that a one-GPU box can run the exact RCCL calls the multi-GPU summary makes
# (all_gather fp64, all_reduce SUM int64 / MIN / MAX fp64, gather of a device tensor with a receive list) before the
# first multi-GPU lease does (tests/test_distributed_gpu.py::test_rccl_first_contact_on_one_gpu).
_FORCE = [os.environ.get("FIVEEQ_FORCE_COLLECTIVES", "") == "1"]


def force_collectives(on=True):
    """Run the exchange's collectives even in a one-rank group (see above).  Returns the previous setting."""
    prev, _FORCE[0] = _FORCE[0], bool(on)
    return prev


def _dist(group):
    """(dist module or None, rank, world, exchange) — `exchange` says whether the collectives run: more than one rank,
    or a one-rank group with force_collectives on."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        world = dist.get_world_size(group)
        return dist, dist.get_rank(group), world, world > 1 or _FORCE[0]
    return None, 0, 1, False


def _comm_tensor(dist, group, x):
    """Collectives run where the backend can reach: device tensors over RCCL ("nccl"), host copies
    over gloo (CPU tests, and single-GPU rehearsals of the multi-process path)."""
    if dist is not None and dist.get_backend(group) == "gloo" and x.is_cuda:
        return x.cpu()
    return x


def merge_moments(parts):
    """parts [W, K, 5] -> [K, 5]: Chan et al. pairwise merge of (count, mean, M2), min/max."""
    out = parts[0].clone()
    for w in range(1, parts.shape[0]):
        b = parts[w]
        na, nb = out[:, 0], b[:, 0]
        n = na + nb
        delta = b[:, 1] - out[:, 1]
        mean = out[:, 1] + delta * nb / n
        m2 = out[:, 2] + b[:, 2] + delta * delta * na * nb / n
        out = torch.stack([n, mean, m2, torch.minimum(out[:, 3], b[:, 3]), torch.maximum(out[:, 4], b[:, 4])], dim=1)
    return out


def linear_positions(n, percentiles):
    """NumPy's 'linear' percentile rule, operation for operation (numpy/lib/_function_base_impl.py: the 'linear' entry of
    _QuantileMethods, _get_indexes, _get_gamma): for each percentile the two order-statistic indices (i0, i1) of an n-member
    row and the weight gamma.  Returns (i0 [P] int64, i1 [P] int64, gamma [P] float64) as NumPy arrays.  With `lerp` below the
    result is np.percentile's BIT FOR BIT (fp64 rows) once the order statistics are exact."""
    q = np.true_divide(np.asarray(percentiles, dtype=np.float64), 100)
    virtual = (n - 1) * q
    prev = np.floor(virtual)
    nxt = prev + 1.0
    above, below = virtual >= n - 1, virtual < 0
    prev[above], nxt[above] = -1.0, -1.0
    prev[below], nxt[below] = 0.0, 0.0
    gamma = virtual - prev                                   # (with prev = -1 above the bounds, like NumPy: both ends are the max then)
    i0, i1 = prev.astype(np.int64) % n, nxt.astype(np.int64) % n
    return i0, i1, gamma


def lerp(a, b, t):
    """numpy's _lerp: a + (b - a) t, and b - (b - a)(1 - t) where t >= 0.5 (NumPy arrays or torch tensors, broadcasting)."""
    d = b - a
    lo = a + d * t
    hi = b - d * (1.0 - t)
    if isinstance(lo, torch.Tensor):
        return torch.where(torch.as_tensor(t >= 0.5, device=lo.device).expand_as(lo), hi, lo)
    return np.where(np.broadcast_to(t >= 0.5, lo.shape), hi, lo)


def percentiles_sorted(xs, percentiles):
    """xs [K, n] sorted along dim 1 -> [K, len(percentiles)], NumPy 'linear' definition (bit for bit in fp64)."""
    i0, i1, gamma = linear_positions(xs.shape[1], percentiles)
    a = xs[:, torch.from_numpy(i0).to(xs.device)]
    b = xs[:, torch.from_numpy(i1).to(xs.device)]
    return lerp(a, b, torch.from_numpy(gamma).to(device=xs.device, dtype=xs.dtype).reshape(1, -1))


def moments_from_sums(sums):
    """sums [n, 5] = (count, sum, sum of squares, min, max) -> dict of [n] tensors."""
    cnt, s1, s2 = sums[:, 0], sums[:, 1], sums[:, 2]
    mean = s1 / cnt
    return {"count": cnt, "mean": mean, "var": (s2 / cnt - mean * mean).clamp_min(0.0), "min": sums[:, 3],
            "max": sums[:, 4]}


def reduce_stats(sums, group=None):
    """All-reduce the per-shard (count, sum, sum^2, min, max) records of every step over the ranks
    (three tiny collectives: SUM on the first three columns, MIN, MAX) and return the ensemble
    moments on every rank.  This is all a run without stored trajectories has to exchange."""
    dist, _, _, exchange = _dist(group)
    sums = _comm_tensor(dist, group, sums).clone()
    if exchange:
        add, mn, mx = sums[:, :3].contiguous(), sums[:, 3].contiguous(), sums[:, 4].contiguous()
        dist.all_reduce(add, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
        sums = torch.cat([add, mn[:, None], mx[:, None]], dim=1)
    return moments_from_sums(sums)


def histogram_percentiles(hist, lo, hi, percentiles=(5.0, 50.0, 95.0), group=None):
    """hist [K, n_bins] int64: this rank's fixed-bin counts (EnsembleEngine.T_histogram) with the SAME
    lo/hi/n_bins on every rank.  One all-reduce(SUM) of K*n_bins counters (24 MB for 750 steps x 4096
    bins) replaces gathering the members; percentiles are read off the cumulative counts with linear
    interpolation inside the bin, so the error is below one bin width w = (hi - lo)/n_bins for
    values inside [lo, hi).  Returns (percentiles [K, P] fp64, total counts [K]) on every rank."""
    dist, _, _, exchange = _dist(group)
    h = _comm_tensor(dist, group, hist).clone()
    if exchange:
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
    n_bins = h.shape[1]
    w = (float(hi) - float(lo)) / n_bins
    cdf = torch.cumsum(h, dim=1).to(torch.float64)
    total = cdf[:, -1]
    cols = []
    for p in percentiles:
        target = (float(p) / 100.0) * total                              # members at or below the percentile
        b = torch.searchsorted(cdf, target[:, None].contiguous()).clamp_(max=n_bins - 1)[:, 0]
        below = torch.where(b > 0, cdf.gather(1, (b - 1).clamp_(min=0)[:, None])[:, 0], torch.zeros_like(total))
        inside = h.gather(1, b[:, None])[:, 0].to(torch.float64)
        frac = torch.where(inside > 0, (target - below) / inside.clamp_min(1.0), torch.zeros_like(total))
        cols.append(float(lo) + (b.to(torch.float64) + frac.clamp(0.0, 1.0)) * w)
    return torch.stack(cols, dim=1), total


SELECT_BINS = 4096


SELECT_CAND_BYTES = 4 << 30      # device summary: rows are summarised in halves while rows x candidates-per-row exceeds this


def _need_device_rows(rows):
    if not _passes_apply(rows):
        raise TypeError(f"rows must be fp32 / fp64 rows on a GPU (got {rows.dtype} on {rows.device}): the summary runs through the "
                        "HIP passes and has no CPU fallback; oracle/summary_host.py restates it in torch ops for the tests")


def exact_percentiles(rows, percentiles, gmin, gmax, n_total, dst=0, group=None, n_bins=SELECT_BINS, stats=None):
    """Exact percentiles (NumPy 'linear' definition) of device rows [K, n_local] over all ranks by histogram selection, with
    the global extrema gmin / gmax [K] and the member count n_total over all ranks handed in (no moments pass): see
    _device_summary.  Returns [K, P] fp64 (a host tensor) on rank `dst`, None elsewhere.  `stats`, if a dict, receives
    bytes_to_root / allreduce_bytes."""
    rows = rows.contiguous()
    _need_device_rows(rows)
    if len(percentiles) < 1:
        raise ValueError("no percentiles asked for")
    _, out = _device_summary(rows, percentiles, dst, group, stats, gmin=gmin, gmax=gmax, n_total=n_total, want_moments=False,
                             n_bins=n_bins)
    return None if out is None else torch.from_numpy(out)


# ---------------------------------------------------------------------------------------------------------------------
# The summary of DEVICE rows: HIP passes + host bookkeeping.
# ---------------------------------------------------------------------------------------------------------------------
def _lib_and_stream(rows):
    """(library, _capi, ctypes, stream) for rows on a GPU.  (tests/test_distributed.py replaces this function to run the host
    side of the summary on CPU tensors against the NumPy restatement of the passes, oracle/summary_passes.py.)"""
    import ctypes

    from . import _capi
    lib = _capi.load()
    return lib, _capi, ctypes, ctypes.c_void_p(torch.cuda.current_stream(rows.device).cuda_stream)


def _passes_apply(rows):
    """The summary of these rows goes through the four passes of the C ABI: fp32 / fp64 rows on a GPU."""
    return rows.is_cuda and rows.dtype in (torch.float32, torch.float64)


def _on(device):
    import contextlib
    return torch.cuda.device(device) if device.type == "cuda" else contextlib.nullcontext()


def _all_reduce(dist, group, x, op):
    """all-reduce of a small device (or host) tensor over the group's backend; gloo takes device tensors through the host."""
    if dist.get_backend(group) == "gloo" and x.is_cuda:
        y = x.cpu()
        dist.all_reduce(y, op=op, group=group)
        x.copy_(y)
    else:
        dist.all_reduce(x, op=op, group=group)
    return x


def _all_gather_np(dist, group, world, arr, device):
    """every rank's NumPy array `arr` (same shape and dtype everywhere) -> array [world, ...], through the group's backend
    (RCCL moves device tensors: they go through `device`, the rank's GPU)."""
    t = torch.from_numpy(np.ascontiguousarray(arr))
    if dist.get_backend(group) != "gloo":
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t, group=group)
    return torch.stack(parts).cpu().numpy()


def device_row_sums(rows, out=None):
    """rows [K, n] on the GPU -> device tensor [K, 4] fp64 = (sum, sum of squares, min, max) per row: ONE pass of
    fiveeq_row_moments_* (16-byte loads, fixed summation order: the same bits every run).  min / max ignore NaNs, the sums
    propagate them.  `out`: a [K, 4] fp64 device tensor to write into."""
    lib, _capi, ctypes, st = _lib_and_stream(rows)
    K, n = rows.shape
    if n == 0:                                         # an empty shard (fewer members than ranks): the neutral record
        if out is None:
            out = torch.empty((K, 4), dtype=torch.float64, device=rows.device)
        out.copy_(torch.tensor([0.0, 0.0, float("inf"), float("-inf")], dtype=torch.float64).expand(K, 4))
        return out
    chunks = int(lib.fiveeq_row_moments_chunks(K, n))
    work = torch.empty((K * chunks + (K if out is None else 0)) * 4, dtype=torch.float64, device=rows.device)
    if out is None:
        out = work[K * chunks * 4:].view(K, 4)
    fn = lib.fiveeq_row_moments_f64 if rows.dtype == torch.float64 else lib.fiveeq_row_moments_f32
    with _on(rows.device):
        _capi.check(lib, fn(K, n, rows.stride(0), ctypes.c_void_p(rows.data_ptr()), ctypes.c_void_p(work.data_ptr()),
                            ctypes.c_void_p(out.data_ptr()), st))
    return out


def _device_summary(rows, percentiles, dst, group, stats, local_sums=None, gmin=None, gmax=None, n_total=None,
                    want_moments=True, n_bins=None):
    """Moments + exact percentiles of device rows [K, n_local] over all ranks.  Returns (mom [K, 5] NumPy fp64 =
    (count, mean, M2, min, max) merged over the ranks — None if want_moments is False —, pct [K, P] NumPy fp64 on rank `dst`
    else None).  local_sums: this rank's [K, 4] (sum, sum^2, min, max) when the caller already has them (the engine's
    in-kernel records); gmin / gmax / n_total: global extrema and member count when the caller already has THOSE
    (exact_percentiles' signature) — then no moments pass runs at all.

    The histogram between the global extrema holds EXACT counts and its bin rule is monotone in the value, so the order
    statistic of global index i is the (i - cdf[b-1])-th smallest member of the bin b with cdf[b-1] <= i < cdf[b].  Hence:
    moments pass -> histogram pass (one rank: ranges straight from the moments, on the device) -> ONE device-to-host copy
    (moments + counts) -> the host marks the bins that hold wanted order statistics and computes each statistic's rank among
    the members of marked bins -> ONE upload -> selection pass (members of marked bins, compacted) -> pick pass (radix
    selection of those ranks) -> ONE copy back.  Several ranks add the exchanges where the text below says so."""
    dist, rank, world, exchange = _dist(group)
    lib, _capi, ctypes, st = _lib_and_stream(rows)
    n_bins = SELECT_BINS if n_bins is None else int(n_bins)
    K, n_local = rows.shape
    P = len(percentiles)
    Q = 2 * P
    dev = rows.device
    f64 = rows.dtype == torch.float64
    w = rows.element_size()
    sfx = "f64" if f64 else "f32"
    ptr = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + off)      # noqa: E731
    if P < 1:
        raise ValueError("no percentiles asked for")
    ld = rows.stride(0)
    host_wire = exchange and dist.get_backend(group) == "gloo"

    # one device buffer for everything that comes back after the histogram: counts [K, n_bins] int64, then sums [K, 4] fp64
    head = torch.zeros(K * n_bins + K * 4, dtype=torch.int64, device=dev)
    counts = head[:K * n_bins].view(K, n_bins)
    sums_dev = head[K * n_bins:].view(torch.float64).view(K, 4)

    # ---- pass 1: moments (unless the caller brought them), and with them the global extrema ---------------------------
    mom = None
    have_sums = gmin is None or want_moments
    if have_sums:
        if local_sums is not None:
            sums_dev.copy_(local_sums.to(device=dev, dtype=torch.float64))
        else:
            device_row_sums(rows, out=sums_dev)
    lo_np = hi_np = None
    if exchange or gmin is not None:
        if have_sums:
            mine = np.concatenate([np.full((K, 1), float(n_local)), sums_dev.cpu().numpy()], axis=1)      # [K, 5]
            parts = _all_gather_np(dist, group, world, mine, dev) if exchange else mine[None]
            cnt, s1 = parts[:, :, 0], parts[:, :, 1]
            keep = cnt[:, 0] > 0                                 # ranks whose shard is empty take no part in the merge
            parts, cnt, s1 = parts[keep], cnt[keep], s1[keep]
            if parts.shape[0] == 0:
                raise ValueError("gather_summary: no members on any rank")
            mean_r = s1 / cnt
            with np.errstate(invalid="ignore", over="ignore"):
                m2_r = np.maximum(parts[:, :, 2] - cnt * mean_r * mean_r, 0.0)
            mom = merge_moments(torch.from_numpy(np.stack([cnt, mean_r, m2_r, parts[:, :, 3], parts[:, :, 4]], axis=2))).numpy()
        if gmin is None:
            lo_np, hi_np, n_tot = mom[:, 3].copy(), mom[:, 4].copy(), int(round(float(mom[0, 0])))
        else:
            lo_np = np.asarray(gmin.detach().cpu() if isinstance(gmin, torch.Tensor) else gmin, dtype=np.float64).reshape(K).copy()
            hi_np = np.asarray(gmax.detach().cpu() if isinstance(gmax, torch.Tensor) else gmax, dtype=np.float64).reshape(K).copy()
            n_tot = int(n_total)
        ranges = torch.from_numpy(np.stack([lo_np, hi_np], axis=1)).to(dev)
    else:
        ranges = sums_dev[:, 2:4].contiguous()          # one rank: the extrema never leave the device before the histogram
        n_tot = n_local

    # ---- pass 2: histograms between the global extrema ------------------------------------------------------------------
    if n_local > 0:
        with _on(dev):
            _capi.check(lib, getattr(lib, f"fiveeq_hist_rows_ranged_{sfx}")(K, n_local, ld, ptr(rows), ptr(ranges), n_bins,
                                                                            ptr(counts), st))
    if exchange:
        _all_reduce(dist, group, counts, dist.ReduceOp.SUM)
    head_np = head.cpu().numpy()                          # THE device-to-host copy of a one-rank summary's first half
    counts_np = head_np[:K * n_bins].reshape(K, n_bins)
    if lo_np is None:
        sums_np = head_np[K * n_bins:].view(np.float64).reshape(K, 4)
        cnt = np.full(K, float(n_local))
        mean_r = sums_np[:, 0] / cnt
        with np.errstate(invalid="ignore", over="ignore"):       # a row with an infinite member: its moments are inf / nan, quietly
            mom = np.stack([cnt, mean_r, np.maximum(sums_np[:, 1] - cnt * mean_r * mean_r, 0.0), sums_np[:, 2], sums_np[:, 3]], axis=1)
        lo_np, hi_np = sums_np[:, 2].copy(), sums_np[:, 3].copy()
    cdf = np.cumsum(counts_np, axis=1)
    # a NaN member has no bin: a row whose counts do not add up to the member count holds one, and np.percentile of it is NaN
    nan_row = cdf[:, -1] != n_tot
    if stats is not None:
        stats["bytes_to_root"], stats["allreduce_bytes"] = 0, (counts.numel() * 8 if exchange else 0)
        stats["bytes_to_root_per_rank"] = [0] * world

    # ---- host: the bins that hold the wanted order statistics, and each statistic's rank among the members of marked bins ---
    if n_tot < 1:
        raise ValueError("gather_summary: no members on any rank")
    i0, i1, frac = linear_positions(n_tot, percentiles)
    want = np.concatenate([i0, i1])                                                     # [2P] global order-statistic indices
    bb = np.stack([np.searchsorted(cdf[k], want, side="right") for k in range(K)]).clip(max=n_bins - 1)     # [K, 2P]
    flat = ~(hi_np > lo_np)                              # a constant row: every member IS the answer, nothing to select
    skip = nan_row | flat
    bb[skip] = 0
    rk = np.arange(K)[:, None]
    marked = np.zeros((K, n_bins), dtype=bool)
    marked[rk, bb] = True
    marked[skip] = False
    below_bin = np.concatenate([np.zeros((K, 1), dtype=np.int64), cdf[:, :-1]], axis=1)                      # members in bins < b
    cand_below = np.concatenate([np.zeros((K, 1), dtype=np.int64), np.cumsum(counts_np * marked, axis=1)[:, :-1]], axis=1)
    ranks = cand_below[rk, bb] + (want[None, :] - below_bin[rk, bb])                    # [K, 2P]
    ranks[skip] = -1
    n_cand = (counts_np * marked).sum(axis=1)                                            # candidates per row over ALL ranks
    if K > 1 and K * int(n_cand.max()) * w > SELECT_CAND_BYTES:
        # one row with heavy ties would size the candidate buffer of EVERY row: summarise the rows in two halves (the decision
        # rests on the all-reduced histogram, so every rank takes it alike); the extrema and the member count are known
        h = K // 2
        half_stats = [None if stats is None else {} for _ in range(2)]
        halves = [_device_summary(rows[a:b], percentiles, dst, group, st_h, gmin=lo_np[a:b], gmax=hi_np[a:b], n_total=n_tot,
                                  want_moments=False, n_bins=n_bins)[1] for (a, b), st_h in zip(((0, h), (h, K)), half_stats)]
        if stats is not None:
            # what travelled is what the halves sent (each half all-reduces its own histogram again: counted too)
            stats["split_rows"] = True
            stats["bytes_to_root"] = sum(st_h["bytes_to_root"] for st_h in half_stats)
            stats["allreduce_bytes"] += sum(st_h["allreduce_bytes"] for st_h in half_stats)
            stats["bytes_to_root_per_rank"] = [sum(v) for v in zip(*(st_h["bytes_to_root_per_rank"] for st_h in half_stats))]
        out = None if halves[0] is None else np.concatenate(halves, axis=0)
        return mom, out
    cap = max(1, int(min(n_local, n_cand.max())))
    words = (n_bins + 31) // 32
    bits = np.zeros((K, words * 32), dtype=np.uint8)
    bits[:, :n_bins] = marked
    binmask = np.packbits(bits.reshape(K, words, 32), axis=2, bitorder="little").view(np.uint32).reshape(K, words)

    # ---- ONE upload: ranks [K, 2P] int64, then the bin masks [K, words] uint32 ------------------------------------------
    up = torch.from_numpy(np.concatenate([ranks.astype(np.int64).reshape(-1).view(np.uint8), binmask.reshape(-1).view(np.uint8)])).to(dev)
    o_mask = K * Q * 8

    # ---- pass 3: selection; pass 4: pick.  tail = [cand_n K (uint64) | picked K*2P (fp64)] ----------------------------------
    select = getattr(lib, f"fiveeq_select_bins_{sfx}")
    pick = getattr(lib, f"fiveeq_select_pick_{sfx}")
    tail = torch.zeros(K + K * Q, dtype=torch.int64, device=dev)
    cand = torch.empty((K, cap), dtype=rows.dtype, device=dev)
    with _on(dev):
        if n_local > 0:
            _capi.check(lib, select(K, n_local, ld, ptr(rows), ptr(ranges), n_bins, ptr(up, o_mask), ptr(cand), cap, ptr(tail), st))
        if not exchange:
            _capi.check(lib, pick(K, 1, cap, ptr(cand), ptr(tail), Q, ptr(up), ptr(tail, K * 8), st))
    tail_np = tail.cpu().numpy()                          # one rank: THE second (and last) device-to-host copy
    cand_n = tail_np[:K]
    tot_c = cand_n
    if exchange:
        # several ranks: the candidates travel to the root (a few thousand values out of the ensemble) and the root's pick
        # pass runs over one segment per rank
        all_n = _all_gather_np(dist, group, world, cand_n, dev)                             # [world, K]
        width = max(int(all_n.max()), 1)
        send = cand[:, :min(width, cap)]
        if send.shape[1] < width:
            send = torch.cat([send, send.new_zeros((K, width - send.shape[1]))], dim=1)
        send = send.contiguous()
        if host_wire:
            send = send.cpu()
        recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, recv, dst=dst, group=group)
        if stats is not None:
            stats["bytes_to_root"] = int(all_n.sum() - all_n[dst].sum()) * w
            stats["bytes_to_root_per_rank"] = [0 if r == dst else int(all_n[r].sum()) * w for r in range(world)]
        if rank != dst:
            return mom, None
        pool = torch.stack(recv, dim=1).to(dev).contiguous()                           # [K, world, width]
        seg_n = torch.from_numpy(np.ascontiguousarray(all_n.T)).to(dev)                # [K, world]
        with _on(dev):
            _capi.check(lib, pick(K, world, width, ptr(pool), ptr(seg_n), Q, ptr(up), ptr(tail, K * 8), st))
        tail_np = tail.cpu().numpy()
        tot_c = all_n.sum(axis=0)
    picked = tail_np[K:].view(np.float64).reshape(K, Q)
    c0, c1 = picked[:, :P], picked[:, P:]
    # the selection pass must have found exactly the members the histogram counted in the marked bins (same rule, same ranges)
    if not np.array_equal(tot_c[~skip], n_cand[~skip]):
        k = int(np.argwhere((tot_c != n_cand) & ~skip)[0, 0])
        raise RuntimeError(f"percentile selection: row {k} has {int(tot_c[k])} candidates where the histogram counts {int(n_cand[k])}")
    with np.errstate(invalid="ignore"):
        good = skip[:, None] | ((c0 <= c1) & (c0 >= lo_np[:, None]) & (c1 <= hi_np[:, None]))
        if not good.all():
            k, j = [int(v) for v in np.argwhere(~good)[0]]
            raise RuntimeError(f"percentile selection lost its order statistic (row {k}, p={percentiles[j]}): ranks "
                               f"{int(ranks[k, j])},{int(ranks[k, j + P])} of {int(tot_c[k])} candidates, values {c0[k, j]}, {c1[k, j]}")
        out = lerp(c0, c1, frac[None, :])
    out = np.where(nan_row[:, None], np.nan, np.where(flat[:, None], np.broadcast_to(lo_np[:, None], (K, P)), out))
    return mom, out


def gather_summary(rows, percentiles=(5.0, 50.0, 95.0), dst=0, group=None, stats=None, local_sums=None):
    """rows [K, n_local] ON THE GPU: this rank's members at K output times.  Collective over `group`.
    Returns on every rank a dict with the merged moments (mean, var, min, max, count; [K] each, fp64, host tensors);
    on rank `dst` it also holds 'percentiles' [K, len(percentiles)] over ALL members (None elsewhere) —
    exact (np.percentile's 'linear' rule bit for bit), found by selection through the four HIP passes (_device_summary).
    `stats` (dict) receives 'bytes_to_root' (candidate members the root received) and 'allreduce_bytes' (the histogram).
    `local_sums` [K, 4] = this rank's (sum, sum of squares, min, max) per row spares the moments pass when the caller already
    has those — the engine's in-kernel records, EnsembleEngine.gather_summary."""
    rows = rows.contiguous()
    _need_device_rows(rows)
    mom_np, pct_np = _device_summary(rows, percentiles, dst, group, stats, local_sums=local_sums)
    mom = torch.from_numpy(mom_np)           # a few numbers per row: they stay on the host
    return {"count": mom[:, 0].clone(), "mean": mom[:, 1].clone(), "var": mom[:, 2] / mom[:, 0], "min": mom[:, 3].clone(),
            "max": mom[:, 4].clone(), "percentiles": None if pct_np is None else torch.from_numpy(pct_np)}
