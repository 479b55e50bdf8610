#!/usr/bin/env python3
"""What does storing T every step cost the VALU-bound fused kernel, and why?  Variants of the row map only."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

LIB = os.environ.get("FIVEEQ_VARIANT_LIB") or None       # e.g. /tmp/fiveeq_variants/lib_nt_store.so
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
dt = torch.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else torch.float64
steps = 128
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=dt)
E = emissions.rcp_like_emissions(750, 3)[300:300 + steps]


def timed(eng, reps=4):
    best = None
    for _ in range(reps):
        eng.reset_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(mode="fused")
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        best = d if best is None else min(best, d)
    return best / steps * 1e6


base = EnsembleEngine(p, N, E, lib_path=LIB, dtype=dt, store_trajectory=False)
print(f"nothing stored                         {timed(base):8.2f} us/step")
del base
for label, out_steps in (("T of every step (128 rows)", None), ("T of every 2nd step", list(range(0, steps, 2))),
                         ("T of every 8th step", list(range(0, steps, 8)))):
    e = EnsembleEngine(p, N, E, lib_path=LIB, dtype=dt, store_concentrations=False, output_steps=out_steps)
    print(f"{label:38s} {timed(e):8.2f} us/step")
    del e
e = EnsembleEngine(p, N, E, lib_path=LIB, dtype=dt, store_concentrations=False, output_steps=[0])
e.drive[:, 7] = 0.0                      # every step overwrites the SAME row: the stores stay on-die
print(f"{'T of every step into ONE row':38s} {timed(e):8.2f} us/step")
del e
e = EnsembleEngine(p, N, E, lib_path=LIB, dtype=dt, store_concentrations=False, output_steps=list(range(8)))
e.drive[:, 7] = torch.arange(steps, device="cuda:0").remainder(8).to(dt)     # an 8-row ring (400 MB at 12.5M fp32)
print(f"{'T of every step into an 8-row ring':38s} {timed(e):8.2f} us/step")
del e
e = EnsembleEngine(p, N, E, dtype=dt)
print(f"{'C and T of every step (4 rows/step)':38s} {timed(e):8.2f} us/step")
