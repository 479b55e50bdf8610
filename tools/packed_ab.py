#!/usr/bin/env python3
"""A/B of the packed fp32 lanes (two members per lane) against the one-member-per-lane kernels on one MI355X:
per-step and fused fp32 kernels at the config-3 size (1M members) and the config-5 shard (12.5M), us per step.
    python3 tools/packed_ab.py [members ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import _capi, emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [1_000_000, 12_500_000]
lib = _capi.load()
E = emissions.rcp_like_emissions(750, 3)


def timed(eng, reps, **kw):
    best = None
    for _ in range(reps + 1):
        eng.reset_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(**kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best / eng.n_steps * 1e6


for N in sizes:
    p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=torch.float32)
    for label, kw_eng, kw_run in (
            ("per_step, trajectories stored", dict(store_trajectory=N <= 2_000_000), dict(mode="per_step")),
            ("per_step, nothing stored", dict(store_trajectory=False), dict(mode="per_step")),
            ("fused, stats on, nothing stored", dict(store_trajectory=False, collect_stats=True), dict(mode="fused")),
            ("fused, stats off, nothing stored", dict(store_trajectory=False), dict(mode="fused")),
            ("fused, T of every step stored", dict(store_concentrations=False) if N <= 2_000_000 else None, dict(mode="fused"))):
        if kw_eng is None:
            continue
        row = []
        for packing in (0, 1):
            lib.fiveeq_set_f32_packing(packing)
            eng = EnsembleEngine(p, N, E, dtype=torch.float32, device="cuda:0", **kw_eng)
            row.append(timed(eng, 2, **kw_run))
            A = eng.bytes_per_member_step(kw_run["mode"])
            eng.close()
            del eng
        print(f"N={N:>9} fp32 {label:<34} scalar {row[0]:8.2f} us/step  packed {row[1]:8.2f} us/step  ({row[1] / row[0] - 1:+.1%})"
              f"   packed: {N / row[1] * 1e6:.3e} member-steps/s, {A * N / row[1] / 1e3 / 8000:.3f} of 8 TB/s on algorithmic bytes",
              flush=True)
lib.fiveeq_set_f32_packing(1)
