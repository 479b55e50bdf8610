"""Multi-GPU side of the engine: member sharding and the end-of-run summary exchange.

Ensemble members never interact, so the ensemble is cut into contiguous member ranges, one per
GPU / process, and time-stepping needs NO communication.  The only exchange is after the last
step: summary statistics of T (or C) at selected output times, over ALL members —
  * moments (count, mean, variance, min, max): each rank reduces its shard, the tiny per-rank
    records are all-gathered and merged with Chan's parallel-variance formula;
  * exact percentiles: each rank's rows are gathered to one root (on an 8-GPU MI355X node the
    root receives over all 7 of its xGMI links at once; 10 MB per rank and output time at
    1.25M members), sorted there, and read off with NumPy's default linear interpolation.
`torch.distributed` backend "nccl" is RCCL on ROCm; the same code runs on CPU tensors over gloo
(tests/test_distributed.py).  The reference has no distributed code at all.
"""
import torch


def shard_bounds(n_total, rank, world):
    """Contiguous, balanced member range [lo, hi) of `rank` (sizes differ by at most one)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside 0..{world - 1}")
    base, extra = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _dist(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(group), dist.get_world_size(group)
    return None, 0, 1


def _comm_tensor(dist, group, x):
    """Collectives run where the backend can reach: device tensors over RCCL ("nccl"), host copies
    over gloo (CPU tests, and single-GPU rehearsals of the multi-process path)."""
    if dist is not None and dist.get_backend(group) == "gloo" and x.is_cuda:
        return x.cpu()
    return x


def local_moments(x):
    """x [K, n] -> [K, 5] = (count, mean, M2, min, max) per row, in fp64."""
    x = x.to(torch.float64)
    n = x.shape[1]
    mean = x.mean(dim=1)
    m2 = ((x - mean[:, None]) ** 2).sum(dim=1)
    cnt = torch.full_like(mean, float(n))
    return torch.stack([cnt, mean, m2, x.min(dim=1).values, x.max(dim=1).values], dim=1)


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


def percentiles_sorted(xs, percentiles):
    """xs [K, n] sorted along dim 1 -> [K, len(percentiles)], NumPy 'linear' definition."""
    n = xs.shape[1]
    cols = []
    for p in percentiles:
        pos = (float(p) / 100.0) * (n - 1)
        lo = int(pos)
        hi = min(lo + 1, n - 1)
        frac = pos - lo
        cols.append(xs[:, lo] + (xs[:, hi] - xs[:, lo]) * frac)
    return torch.stack(cols, dim=1)


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
    dist, _, world = _dist(group)
    sums = _comm_tensor(dist, group, sums).clone()
    if world > 1:
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
    dist, _, world = _dist(group)
    h = _comm_tensor(dist, group, hist).clone()
    if world > 1:
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


def gather_summary(rows, percentiles=(5.0, 50.0, 95.0), dst=0, group=None):
    """rows [K, n_local]: this rank's members at K output times.  Collective over `group`.
    Returns on every rank a dict with the merged moments (mean, var, min, max, count; [K] each, fp64);
    on rank `dst` it also holds 'percentiles' [K, len(percentiles)] over ALL members (None elsewhere)."""
    dist, rank, world = _dist(group)
    rows = _comm_tensor(dist, group, rows.contiguous())
    K, n_local = rows.shape
    mom = local_moments(rows)
    if world > 1:
        parts = [torch.empty_like(mom) for _ in range(world)]
        dist.all_gather(parts, mom, group=group)
        mom = merge_moments(torch.stack(parts))
        # shard sizes may differ by one: pad to the largest with +inf (sorts to the end, then dropped)
        sizes = [torch.zeros(1, dtype=torch.int64, device=rows.device) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([n_local], dtype=torch.int64, device=rows.device), group=group)
        sizes = [int(s.item()) for s in sizes]
        n_max = max(sizes)
        send = rows if n_local == n_max else torch.cat(
            [rows, torch.full((K, n_max - n_local), float("inf"), dtype=rows.dtype, device=rows.device)], dim=1)
        recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, recv, dst=dst, group=group)
        if rank == dst:
            allrows = torch.cat([r[:, :s] for r, s in zip(recv, sizes)], dim=1)
        else:
            allrows = None
    else:
        allrows = rows
    out = {"count": mom[:, 0], "mean": mom[:, 1], "var": mom[:, 2] / mom[:, 0], "min": mom[:, 3], "max": mom[:, 4],
           "percentiles": None}
    if allrows is not None:
        xs, _ = torch.sort(allrows.to(torch.float64), dim=1)
        out["percentiles"] = percentiles_sorted(xs, percentiles)
    return out
