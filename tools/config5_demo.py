#!/usr/bin/env python3
"""BASELINE configs[4], one GPU's shard, end to end the way such a job is run: rank 3 of 8's 12.5M members of the
100M-member Latin hypercube (drawn on the device), fp32, time-fused, NO trajectory stored: per-step moments and
750 x 4096-bin histograms of T are accumulated while the model runs (streamed pipeline), three years are kept for
exact percentiles by selection."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.distributed import gather_summary, histogram_percentiles, shard_bounds  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N_TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
MODE = sys.argv[2] if len(sys.argv) > 2 else "fused"
lo_m, hi_m = shard_bounds(N_TOTAL, 3, 8)
N = hi_m - lo_m
torch.cuda.init()
torch.zeros(1, device="cuda:0")
t0 = time.perf_counter()
p = params.sample_ensemble_shard(params.default_params("multigas"), N_TOTAL, lo_m, hi_m, device="cuda:0",
                                 dtype=torch.float32)
E = emissions.rcp_like_emissions(750, 3)
torch.cuda.synchronize()
t1 = time.perf_counter()
LO, HI, NB = -2.0, 12.0, 4096
eng = EnsembleEngine(p, N, E, dtype=torch.float32, device="cuda:0", output_steps=[249, 499, 749],
                     store_concentrations=False, collect_stats=True, hist=(LO, HI, NB))
torch.cuda.synchronize()
t2 = time.perf_counter()
eng.run(mode=MODE)                                   # first pass: allocates the ring, loads the code objects
torch.cuda.synchronize()
t3 = time.perf_counter()
eng.reset_state()
torch.cuda.synchronize()
t3a = time.perf_counter()
eng.run(mode=MODE)                                   # steady state
torch.cuda.synchronize()
t3b = time.perf_counter()
st = eng.stats()
t3c = time.perf_counter()
pct, tot = histogram_percentiles(eng.T_hist, LO, HI, (5.0, 50.0, 95.0))
torch.cuda.synchronize()
t4 = time.perf_counter()
ex = gather_summary(eng.T, percentiles=(5.0, 50.0, 95.0))["percentiles"]
torch.cuda.synchronize()
t5 = time.perf_counter()
mem = torch.cuda.max_memory_allocated() / 1e9
print(f"members {N} (rank 3 of 8 of {N_TOTAL}), fp32, 750 steps, 3 gases, mode {MODE}")
print(f"  GPU : shard of the Latin hypercube drawn on the device {t1 - t0:.3f} s; allocation {t2 - t1:.3f} s; "
      f"peak device memory {mem:.2f} GB (a stored T[750][N] alone would be {750 * N * 4 / 1e9:.1f} GB)")
print(f"  GPU : run incl. per-step moments and in-loop 750 x {NB}-bin histograms: first pass {t3 - t2:.3f} s, repeated "
      f"{t3b - t3a:.3f} s = {N * 750 / (t3b - t3a):.3e} member-timesteps/s")
print(f"  GPU : all-step percentiles from the histograms {t4 - t3c:.4f} s; exact percentiles of 3 stored years by "
      f"selection {t5 - t4:.4f} s")
for row, t in enumerate((249, 499, 749)):
    print(f"  step {t}: mean {st['mean'][t].item():.4f} K, p05/p50/p95 histogram "
          f"{pct[t, 0].item():.4f} / {pct[t, 1].item():.4f} / {pct[t, 2].item():.4f}   exact "
          f"{ex[row, 0].item():.4f} / {ex[row, 1].item():.4f} / {ex[row, 2].item():.4f}")
print(f"  bin width {(HI - LO) / NB:.5f} K; counts per step {int(tot.min().item())}..{int(tot.max().item())}")
