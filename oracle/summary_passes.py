"""CPU restatement of the four end-of-run-summary passes of the C ABI (include/fiveeq.h, "END-OF-RUN SUMMARY":
fiveeq_row_moments_*, fiveeq_hist_rows_ranged_*, fiveeq_select_bins_*, fiveeq_select_pick_*) in NumPy, behind the same
pointer-and-size signatures.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): tests/test_distributed.py hands this object to
fiveeqscm_amd.distributed in place of the HIP library so that the host side of the summary — which bins are marked, the
rank bookkeeping, the packed uploads, the multi-rank exchange over gloo — runs on CPU tensors in the container that has no
GPU.  The product never imports it; the engine has no CPU path.

The reference (stujen/fiveEqSCM @ v0) has no counterpart: no summary statistics, no percentiles.
"""
import ctypes

import numpy as np

_CT = {np.float64: ctypes.c_double, np.float32: ctypes.c_float, np.int64: ctypes.c_int64, np.uint64: ctypes.c_uint64,
       np.uint32: ctypes.c_uint32}


def _view(ptr, dtype, count):
    """NumPy view of `count` elements of `dtype` at the raw address behind a ctypes.c_void_p (or an int)."""
    addr = ptr.value if isinstance(ptr, ctypes.c_void_p) else int(ptr)
    if count == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array((_CT[dtype] * count).from_address(addr))


def _rows(ptr, dtype, n_rows, n, ld):
    return _view(ptr, dtype, (n_rows - 1) * ld + n if n_rows else 0), [slice(k * ld, k * ld + n) for k in range(n_rows)]


def bin_rule(x, lo, hi, n_bins, dtype):
    """THE BIN RULE of include/fiveeq.h for rows of `dtype`; NaN -> -1 (no bin).  fp32 rows: one fp32 FMA (restated in fp64
    and rounded to fp32 once), fp64 rows: the fp64 formula.  Both monotone in x, which is all the selection needs."""
    inv_w = n_bins / (hi - lo) if hi > lo else 0.0
    with np.errstate(invalid="ignore", over="ignore"):
        if dtype == np.float32:
            pos = (x.astype(np.float64) * np.float64(np.float32(inv_w)) + np.float64(np.float32(-lo * inv_w))).astype(np.float32)
        else:
            pos = (x - lo) * inv_w
        b = np.trunc(np.clip(np.nan_to_num(pos, nan=0.0), 0.0, n_bins - 1)).astype(np.int64)
    return np.where(np.isnan(x), -1, b)


class SummaryPasses:
    """The four passes with the C ABI's signatures; `stream` is ignored.  Every function returns 0."""

    def fiveeq_row_moments_chunks(self, n_rows, n):
        return 1 if n_rows > 0 and n > 0 else 0

    def _moments(self, dtype, n_rows, n, ld, rows, partial, moments, stream):
        flat, sl = _rows(rows, dtype, n_rows, n, ld)
        out = _view(moments, np.float64, n_rows * 4).reshape(n_rows, 4)
        for k in range(n_rows):
            x = flat[sl[k]].astype(np.float64)
            with np.errstate(invalid="ignore"):
                out[k] = (x.sum(), (x * x).sum(), np.nanmin(x) if not np.isnan(x).all() else np.inf,
                          np.nanmax(x) if not np.isnan(x).all() else -np.inf)
        return 0

    def _hist(self, dtype, n_rows, n, ld, rows, ranges, n_bins, hist, stream):
        flat, sl = _rows(rows, dtype, n_rows, n, ld)
        rg = _view(ranges, np.float64, n_rows * 2).reshape(n_rows, 2)
        h = _view(hist, np.int64, n_rows * n_bins).reshape(n_rows, n_bins)
        for k in range(n_rows):
            b = bin_rule(flat[sl[k]], rg[k, 0], rg[k, 1], n_bins, dtype)
            h[k] += np.bincount(b[b >= 0], minlength=n_bins)
        return 0

    def _select(self, dtype, n_rows, n, ld, rows, ranges, n_bins, binmask, cand, cap, cand_n, stream):
        flat, sl = _rows(rows, dtype, n_rows, n, ld)
        rg = _view(ranges, np.float64, n_rows * 2).reshape(n_rows, 2)
        words = (n_bins + 31) // 32
        mask = _view(binmask, np.uint32, n_rows * words).reshape(n_rows, words)
        out = _view(cand, dtype, n_rows * cap).reshape(n_rows, cap) if cap else None
        cn = _view(cand_n, np.uint64, n_rows)
        for k in range(n_rows):
            bits = np.unpackbits(mask[k].view(np.uint8), bitorder="little")[:n_bins].astype(bool)
            x = flat[sl[k]]
            b = bin_rule(x, rg[k, 0], rg[k, 1], n_bins, dtype)
            pick = x[(b >= 0) & bits[np.maximum(b, 0)]]
            pick = pick[::-1]                                   # any order: the kernel's is not the row's either
            start = int(cn[k])
            cn[k] += np.uint64(pick.size)
            room = max(0, min(pick.size, cap - start))
            if room:
                out[k, start:start + room] = pick[:room]
        return 0

    def _pick(self, dtype, n_rows, n_seg, width, pool, seg_n, n_targets, ranks, picked, stream):
        p = _view(pool, dtype, n_rows * n_seg * width).reshape(n_rows, n_seg, width)
        sn = _view(seg_n, np.uint64, n_rows * n_seg).reshape(n_rows, n_seg)
        rk = _view(ranks, np.int64, n_rows * n_targets).reshape(n_rows, n_targets)
        out = _view(picked, np.float64, n_rows * n_targets).reshape(n_rows, n_targets)
        for k in range(n_rows):
            c = np.sort(np.concatenate([p[k, g, :int(sn[k, g])] for g in range(n_seg)]).astype(np.float64))
            for q in range(n_targets):
                out[k, q] = c[rk[k, q]] if 0 <= rk[k, q] < c.size else np.nan
        return 0

    def fiveeq_last_error(self):
        return b""


for _name, _dt in (("f64", np.float64), ("f32", np.float32)):
    for _fn, _impl in (("fiveeq_row_moments", "_moments"), ("fiveeq_hist_rows_ranged", "_hist"),
                       ("fiveeq_select_bins", "_select"), ("fiveeq_select_pick", "_pick")):
        setattr(SummaryPasses, f"{_fn}_{_name}",
                (lambda impl, dt: lambda self, *a: getattr(self, impl)(dt, *a))(_impl, _dt))
