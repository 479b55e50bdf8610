#!/usr/bin/env python3
"""Exact percentiles of EVERY stored step at once: gather_summary on the whole T[750][N] of a 1M-member fp64 run (K = 750 rows),
against np.percentile on a few rows and against the one-bin-accurate histogram percentiles.   python3 tools/summary_all_steps.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.distributed import gather_summary  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0")
eng = EnsembleEngine(p, N, emissions.rcp_like_emissions(750, 3), device="cuda:0", store_concentrations=False)
eng.run(mode="fused")
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    s = gather_summary(eng.T[1:], (5.0, 50.0, 95.0))           # (row 0 is the first step; T[0] of a zero start is nearly constant)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"{N} members x {eng.T.shape[0] - 1} rows fp64: exact p05 / p50 / p95 of every row in {dt * 1e3:.1f} ms "
      f"({dt / (eng.T.shape[0] - 1) * 1e6:.0f} us per row)")
for k in (0, 300, 748):
    want = np.percentile(eng.T[1 + k].cpu().numpy(), (5.0, 50.0, 95.0))
    assert np.array_equal(s["percentiles"][k].numpy(), want), k
print("rows 1, 301, 749 equal np.percentile bit for bit")
