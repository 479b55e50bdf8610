#!/usr/bin/env python3
"""BASELINE configs[4], one GPU's shard, end to end the way such a job is run: 12.5M members, fp32,
time-fused, T stored for all 750 steps (37.5 GB; concentrations not stored), per-step moments on the
device, and every step's 5/50/95th percentiles from fixed-bin histograms of the stored rows."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.distributed import histogram_percentiles  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
t0 = time.perf_counter()
torch.cuda.init()
torch.zeros(1, device="cuda:0")
t0 = time.perf_counter()
p = params.sample_ensemble_device(params.default_params("multigas"), N, "cuda:0")   # LHS drawn on the GPU
E = emissions.rcp_like_emissions(750, 3)
torch.cuda.synchronize()
t1 = time.perf_counter()
eng = EnsembleEngine(p, N, E, dtype=torch.float32, device="cuda:0", store_concentrations=False, collect_stats=True)
torch.cuda.synchronize()
t2 = time.perf_counter()
eng.run(mode="fused")
torch.cuda.synchronize()
t3 = time.perf_counter()
st = eng.stats()
lo, hi = float(st["min"].min()) - 1e-3, float(st["max"].max()) + 1e-3
hist = eng.T_histogram(lo, hi, 4096)
pct, tot = histogram_percentiles(hist, lo, hi, (5.0, 50.0, 95.0))
torch.cuda.synchronize()
t4 = time.perf_counter()
exact = torch.sort(eng.T[749].double()).values
ex = [exact[int(f * (N - 1))].item() for f in (0.05, 0.5, 0.95)]
print(f"members {N}, fp32, 750 steps, 3 gases")
print(f"  GPU : Latin-hypercube parameters drawn on the device {t1 - t0:.3f} s; allocation {t2 - t1:.3f} s")
print(f"  GPU : fused run {t3 - t2:.3f} s = {N * 750 / (t3 - t2):.3e} member-timesteps/s")
print(f"  GPU : moments + 750 x 4096-bin histograms + percentiles {t4 - t3:.3f} s")
for t in (249, 499, 749):
    print(f"  step {t}: mean {st['mean'][t].item():.4f} K, p05/p50/p95 from histogram "
          f"{pct[t, 0].item():.4f} / {pct[t, 1].item():.4f} / {pct[t, 2].item():.4f}")
print(f"  step 749 exact p05/p50/p95: {ex[0]:.4f} / {ex[1]:.4f} / {ex[2]:.4f}   (bin width {(hi - lo) / 4096:.5f} K)")
