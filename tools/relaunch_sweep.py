#!/usr/bin/env python3
"""One fused launch against the same kernel relaunched every K steps (mode 'ksteps'), over ensemble sizes: a SIMD serves its oldest
wave first, so a long launch with few rounds of waves ends in a long tail; relaunching resets the ages.
    python3 tools/relaunch_sweep.py [f64|f32]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

dt = torch.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else torch.float64
E = emissions.rcp_like_emissions(750, 3)
print(f"{'f32' if dt == torch.float32 else 'f64'}, 750 steps, statistics on, nothing stored; us/step")
for N in (100_000, 250_000, 500_000, 1_000_000, 1_250_000, 2_000_000, 4_000_000, 8_000_000):
    p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=dt)
    eng = EnsembleEngine(p, N, E, dtype=dt, store_trajectory=False, collect_stats=True)
    out = []
    for mode, k in (("fused", None), ("ksteps", 16), ("ksteps", 32), ("ksteps", 64), ("ksteps", 128), ("ksteps", 250)):
        best = None
        for _ in range(3):
            eng.reset_state()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run(mode=mode, k_steps=k)
            torch.cuda.synchronize()
            d = time.perf_counter() - t0
            best = d if best is None else min(best, d)
        out.append(f"{'one launch' if k is None else 'K=%d' % k} {best / 750 * 1e6:7.2f}")
    print(f"{N:8d}  " + "   ".join(out), flush=True)
    eng.close()
    del eng, p
