#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc SQ_* passes of the time-fused family: launches of STEPS steps each (the reducer divides
per-wave counts by STEPS) of the plain fused kernel — and, for a single-gas layout, of the small-ensemble kernel at 4 and 1
lanes per member.
    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES ... -- python3 tools/pmc_workload_fused.py [members] [f64|f32] [steps] [kind] [small|comp]
(small, three gases: one member per octet of lanes and one per lane; comp: the compensated fp32 form of the fused kernel)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.float64
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 96
kind = sys.argv[4] if len(sys.argv) > 4 else "multigas"
small = len(sys.argv) > 5 and sys.argv[5] == "small"
comp = len(sys.argv) > 5 and sys.argv[5] == "comp"
p = params.sample_ensemble_shard(params.default_params(kind), N, device="cuda:0", dtype=dt)
E = emissions.rcp_like_emissions(max(750, 250 + STEPS), 3 if kind == "multigas" else 1)[250:250 + STEPS]
assert E.shape[0] == STEPS
if small:
    for lanes in ((4, 1) if kind == "co2" else (8, 1)):
        eng = EnsembleEngine(p, N, E, dtype=dt, device="cuda:0", small_lanes=lanes)       # trajectories stored, like config 2
        for _ in range(3):
            eng.reset_state()
            eng.run(mode="small")
            torch.cuda.synchronize()
        eng.close()
else:
    eng = EnsembleEngine(p, N, E, dtype=dt, device="cuda:0", store_trajectory=False, collect_stats=True, compensated=comp)
    for _ in range(2):
        eng.reset_state()
        eng.run(mode="fused")
        torch.cuda.synchronize()
print("pmc fused workload done", N, dt, STEPS, kind, "small" if small else "fused")
