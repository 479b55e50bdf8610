#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc SQ_* passes of the time-fused family: fused and tiled launches of STEPS steps
each (the reducer divides per-wave counts by STEPS).
    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES ... -- python3 tools/pmc_workload_fused.py [members] [f64|f32] [steps] [tile_k] [kind]"""
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
kind = sys.argv[5] if len(sys.argv) > 5 else "multigas"
p = params.sample_ensemble_shard(params.default_params(kind), N, device="cuda:0", dtype=dt)
E = emissions.rcp_like_emissions(750, 3 if kind == "multigas" else 1)[250:250 + STEPS]
eng = EnsembleEngine(p, N, E, dtype=dt, device="cuda:0", store_trajectory=False, collect_stats=True,
                     hist=(-2.0, 12.0, 4096))
eng._wave_stats()            # the raw C entry below bypasses run(): allocate the wave records first, so that the fused
for _ in range(2):           # counters include the in-kernel statistics exactly like the tiled ones do
    eng.reset_state()
    fn = getattr(eng.lib, f"fiveeq_run_fused_{eng._sfx}")          # the plain fused kernel (no histogram pipeline)
    assert fn(*eng._run_args(0, STEPS), eng._stream()) == 0
    torch.cuda.synchronize()
# the tiled kernel as ONE launch of min(STEPS, tile_steps) is not comparable; run it with k_steps dividing STEPS
k = int(sys.argv[4]) if len(sys.argv) > 4 else 8
assert STEPS % k == 0 and k <= eng.tile_steps()
eng.reset_state()
eng.run(mode="tiled", k_steps=k)
torch.cuda.synchronize()
print("pmc fused workload done", N, dt, STEPS, "tile k_steps", k)
