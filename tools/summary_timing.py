#!/usr/bin/env python3
"""End-of-run summary (moments + exact percentiles by selection) on one GPU: warm time vs ensemble size and dtype,
and the device time of each HIP pass."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd.distributed import exact_percentiles, gather_summary  # noqa: E402

for n, dt in ((1_000_000, torch.float64), (1_250_000, torch.float64), (12_500_000, torch.float32), (12_500_000, torch.float64)):
    x = torch.randn((3, n), device="cuda:0", dtype=dt) * 0.7 + 2.0
    gather_summary(x)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        out = gather_summary(x)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    mn, mx = x.min(1).values.double(), x.max(1).values.double()
    exact_percentiles(x, (5.0, 50.0, 95.0), mn, mx, n)
    torch.cuda.synchronize()
    t_sel = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        sel = exact_percentiles(x, (5.0, 50.0, 95.0), mn, mx, n)
        torch.cuda.synchronize()
        t_sel = min(t_sel, time.perf_counter() - t0)
    assert torch.allclose(sel, out["percentiles"], rtol=1e-13)
    t_sort = t_sel
    want = np.percentile(x[0].double().cpu().numpy(), (5.0, 50.0, 95.0))
    assert np.allclose(out["percentiles"][0].cpu().numpy(), want, rtol=1e-13)
    print(f"{n:9d} members x 3 rows {str(dt):14s}: gather_summary (moments + histogram + selection, HIP passes) "
          f"{best * 1e3:7.3f} ms;   exact_percentiles alone (extrema given) {t_sort * 1e3:7.3f} ms")

# device time of the passes alone (HIP events; 3 x 12.5M fp32)
from fiveeqscm_amd.distributed import device_row_sums  # noqa: E402
x = torch.randn((3, 12_500_000), device="cuda:0", dtype=torch.float32) * 0.7 + 2.0
for name, fn in (("row_moments", lambda: device_row_sums(x)),):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name}: {us:.1f} us per call on 3 x 12.5M fp32 = {x.numel() * 4 / us / 1e6:.2f} TB/s")
