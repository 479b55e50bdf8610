#!/usr/bin/env python3
"""Workload for the PMC passes that compare the persistent time-tiled kernel with the fused kernel relaunched every K steps
(same state round trip every K steps, non-persistent workgroups), no histogram, statistics on.
    rocprofv3 --pmc <counters> --kernel-trace -- python3 tools/pmc_workload_tile.py [members] [K] [steps] [f32|f64] [packing 0|1]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 96
dt = torch.float64 if (len(sys.argv) > 4 and sys.argv[4] == "f64") else torch.float32
packing = int(sys.argv[5]) if len(sys.argv) > 5 else 0
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=dt)
E = emissions.rcp_like_emissions(750, 3)[250:250 + STEPS]
eng = EnsembleEngine(p, N, E, dtype=dt, device="cuda:0", store_trajectory=False, collect_stats=True)
eng.lib.fiveeq_set_f32_packing(packing)        # 0: the fused kernel in its one-member-per-lane form, like the tile kernel
for mode in ("ksteps", "tiled", "ksteps", "tiled"):
    eng.reset_state()
    eng.run(mode=mode, k_steps=K)
    torch.cuda.synchronize()
print("pmc tile workload done", N, K, STEPS, dt, "packing", packing)
