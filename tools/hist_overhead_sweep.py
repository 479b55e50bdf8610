#!/usr/bin/env python3
"""Cost of the streamed in-loop histograms (engine mode 'fused' with hist=) against the stats-only fused run, over
ensemble size and precision.  No trajectory stored, stats on, 4096 bins, default ring."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

steps = 750
E = emissions.rcp_like_emissions(steps, 3)


def timed(eng, reps=3):
    best = None
    for _ in range(reps + 1):
        eng.reset_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(mode="fused")
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        best = d if best is None else min(best, d)
    return best / steps * 1e6


print(f"{'dtype':5s} {'members':>9s} {'fused, stats only':>18s} {'+ streamed hist':>16s} {'overhead':>9s} {'ring steps':>10s} "
      f"{'resident WG generations':>24s}")
for dt, sizes in ((torch.float64, (250_000, 1_000_000, 2_000_000, 4_000_000, 8_000_000)),
                  (torch.float32, (500_000, 1_000_000, 2_000_000, 4_000_000, 8_000_000, 12_500_000))):
    for N in sizes:
        p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=dt)
        a = EnsembleEngine(p, N, E, dtype=dt, store_trajectory=False, collect_stats=True)
        t0 = timed(a)
        del a
        b = EnsembleEngine(p, N, E, dtype=dt, store_trajectory=False, collect_stats=True, hist=(-2.0, 12.0, 4096))
        t1 = timed(b)
        S = b.hist_ring_steps
        del b
        per_gen = 256 * (6 if dt == torch.float32 else 4) * 256
        print(f"{'f64' if dt == torch.float64 else 'f32':5s} {N:9d} {t0:15.2f} us {t1:13.2f} us {t1 / t0 - 1:+8.1%} {S:10d} "
              f"{N / per_gen:24.1f}", flush=True)
        torch.cuda.empty_cache()
