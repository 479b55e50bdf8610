#!/usr/bin/env python3
"""A/B of workgroup shapes (builds with -DFIVEEQ_BLOCK=... / -DFIVEEQ_TILE_BLOCK=..., see tools/variant_ab.py): is the cost of
the time-tiled kernel's shape its 1024-thread workgroup?  fp32, config-5 shard, statistics on, nothing stored.

    python3 tools/block_shape_ab.py default /tmp/fiveeq_variants/libfiveeq_B1024.so ...
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N = int(os.environ.get("N_BIG", 12_500_000))
STEPS = 750
STATS = os.environ.get("STATS", "1") == "1"
E = emissions.rcp_like_emissions(STEPS, 3)
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=torch.float32)


def timed(eng, **kw):
    best = None
    for _ in range(3):
        eng.reset_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(**kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best / STEPS * 1e6


print(f"{N} members f32, {STEPS} steps, statistics {'on' if STATS else 'off'}, nothing stored; us/step")
for path in sys.argv[1:]:
    lib_path = None if path == "default" else path
    eng = EnsembleEngine(p, N, E, dtype=torch.float32, store_trajectory=False, collect_stats=STATS, lib_path=lib_path)
    row = [f"{os.path.basename(path):24s}", f"fused {timed(eng, mode='fused'):6.2f}", f"relaunched every 32 {timed(eng, mode='ksteps', k_steps=32):6.2f}"]
    for k in (32, 8):
        row.append(f"tiled K={k} no hist {timed(eng, mode='tiled', k_steps=k):6.2f}")
    eng.close()
    del eng
    if os.environ.get("WITH_HIST", "1") == "1":
        eng = EnsembleEngine(p, N, E, dtype=torch.float32, store_trajectory=False, collect_stats=True, hist=(-2.0, 12.0, 4096),
                             lib_path=lib_path)
        k = eng.tile_steps()
        row.append(f"tiled K={k} 4096 bins {timed(eng, mode='tiled'):6.2f}")
        assert eng.T_hist.sum(1).min().item() == N
        eng.close()
        del eng
    print("  ".join(row), flush=True)
