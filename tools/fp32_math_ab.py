#!/usr/bin/env python3
"""fp32 against 50-digit arithmetic over the 24 golden members, 750 steps, for both fp32 math settings (include/fiveeq.h,
f32_math), and what the accurate setting costs on the time-fused kernel.    python3 tools/fp32_math_ab.py"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import fiveeq_cases as cases  # noqa: E402
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

with open(os.path.join(ROOT, "tests", "golden", "fiveeq_mp_reference.json")) as fh:
    ref = json.load(fh)
print("worst relative error against 50-digit arithmetic over the 24 golden members x 750 steps (stored steps):")
for math in ("fast", "accurate"):
    for kind in ("co2", "multigas"):
        p, N = cases.members(kind)
        eng = EnsembleEngine(p, N, cases.scenario(kind), dtype=torch.float32, device="cuda:0", output_steps=cases.STEPS, fp32_math=math)
        eng.run(mode="fused")
        torch.cuda.synchronize()
        C, T = eng.C.double().cpu().numpy(), eng.T.double().cpu().numpy()
        eC = eT = 0.0
        for i, m in enumerate(ref["members"]):
            C_mp = np.array([[float(v) for v in row] for row in ref["cases"][kind]["C"][i]])
            T_mp = np.array([float(v) for v in ref["cases"][kind]["T"][i]])
            eC = max(eC, float((np.abs(C[:, :, m] - C_mp) / np.abs(C_mp)).max()))
            eT = max(eT, float((np.abs(T[:, m] - T_mp) / (np.abs(T_mp) + 1e-2)).max()))
        print(f"  {math:9s} {kind:9s} C {eC:.2e}   T {eT:.2e} (floor 1e-2 K)")
N = 4_000_000
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=torch.float32)
E = emissions.rcp_like_emissions(750, 3)
for math in ("fast", "accurate"):
    for mode in ("fused", "per_step"):
        eng = EnsembleEngine(p, N, E, dtype=torch.float32, device="cuda:0", store_trajectory=False, fp32_math=math)
        best = None
        for _ in range(3):
            eng.reset_state()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run(mode=mode)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        print(f"  {N} members fp32, no trajectory, {mode:8s} {math:9s} {best / 750 * 1e6:8.2f} us/step")
        eng.close()
