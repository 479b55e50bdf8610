#!/usr/bin/env python3
"""The summary's histogram pass (fiveeq_hist_rows_ranged_*) on 3 x 12.5M values against the number of bins: what each workgroup's
zero + flush of its LDS histogram costs beside the read.     python3 tools/hist_flush_probe.py"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fiveeqscm_amd import _capi
lib = _capi.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())
for dt, sfx in ((torch.float32, "f32"), (torch.float64, "f64")):
    x = torch.randn((3, 12_500_000), device="cuda", dtype=dt) * 0.7 + 2.0
    rg = torch.stack([x.min(1).values.double(), x.max(1).values.double()], 1).contiguous()
    for nb in (4096, 1024, 256, 16):
        h = torch.zeros((3, nb), dtype=torch.int64, device="cuda")
        fn = getattr(lib, f"fiveeq_hist_rows_ranged_{sfx}")
        for _ in range(3):
            fn(3, x.shape[1], x.shape[1], p(x), p(rg), nb, p(h), None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn(3, x.shape[1], x.shape[1], p(x), p(rg), nb, p(h), None)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"{sfx} 3 x 12.5M, {nb:5d} bins: {us:7.1f} us  {x.numel() * x.element_size() / us / 1e6:.2f} TB/s")
