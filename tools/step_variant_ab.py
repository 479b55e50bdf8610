#!/usr/bin/env python3
"""Per-step kernel timing across builds of the library (occupancy variants etc.): us per step, best of 3 passes.
    python3 tools/step_variant_ab.py default /tmp/fiveeq_variants/libfiveeq_X.so ..."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

E = emissions.rcp_like_emissions(750, 3)
cases = [(1_000_000, torch.float64), (1_000_000, torch.float32), (500_000, torch.float64), (2_000_000, torch.float64)]
ps = {(N, dt): params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=dt) for N, dt in cases}
for path in sys.argv[1:]:
    lib_path = None if path == "default" else path
    out = [os.path.basename(path)]
    for N, dt in cases:
        eng = EnsembleEngine(ps[(N, dt)], N, E, dtype=dt, device="cuda:0", lib_path=lib_path)
        best = None
        for _ in range(4):
            eng.reset_state()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run(mode="per_step")
            torch.cuda.synchronize()
            d = time.perf_counter() - t0
            best = d if best is None else min(best, d)
        out.append(f"{N} {'f64' if dt == torch.float64 else 'f32'}: {best / 750 * 1e6:.2f}")
        eng.close()
        del eng
    print(" | ".join(out), flush=True)
