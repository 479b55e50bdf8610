#!/usr/bin/env python3
"""Workload for the PMC passes that look at what the fused kernel's trajectory stores cost: ONE fused launch of STEPS
steps over N fp64 members with (traj = 1) or without (traj = 0) the C/T rows stored.
    rocprofv3 --pmc <counters> --kernel-trace -- python3 tools/pmc_workload_store.py <traj 0|1|2> [members] [steps] [f64|f32]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

traj = int(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 250
dt = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else torch.float64
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=dt)
E = emissions.rcp_like_emissions(750, 3)[250:250 + STEPS]
eng = EnsembleEngine(p, N, E, dtype=dt, device="cuda:0", store_trajectory=bool(traj), store_concentrations=traj == 1)   # traj = 2: T rows only
for _ in range(3):
    eng.reset_state()
    eng.run(mode="fused")
    torch.cuda.synchronize()
print("pmc store workload done", N, STEPS, "traj", traj)
