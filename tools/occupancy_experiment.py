#!/usr/bin/env python3
"""A/B: what the time-tiled kernel's shape costs against the fused kernel.  Variant libraries are built with
`make -C fiveeqscm_amd/csrc OUT=/tmp/fiveeq_variants/<name>.so EXTRA=-D...` (see profiles/r02/ab_variants.txt)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
dt = torch.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else torch.float64
steps = 300
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=dt)
E = emissions.rcp_like_emissions(steps, 3)


def timed(eng, **kw):
    best = None
    for _ in range(4):
        eng.reset_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(**kw)
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        best = d if best is None else min(best, d)
    return best / steps * 1e6


VARIANTS = "/tmp/fiveeq_variants"        # built on the GPU box: make -C fiveeqscm_amd/csrc OUT=... EXTRA=-D...
for name in [None] + (sorted(os.listdir(VARIANTS)) if os.path.isdir(VARIANTS) else []):
    path = None if name is None else os.path.join(VARIANTS, name)
    if name is not None and not name.endswith(".so"):
        continue
    for stats in (True, False):
        eng = EnsembleEngine(p, N, E, dtype=dt, store_trajectory=False, collect_stats=stats, lib_path=path)
        row = f"{name or 'default':24s} stats={int(stats)}  fused {timed(eng, mode='fused'):8.2f}"
        for k in (32, 8):
            row += f"  tiled K={k} {timed(eng, mode='tiled', k_steps=k):8.2f}"
        print(row + "  us/step", flush=True)
        del eng
