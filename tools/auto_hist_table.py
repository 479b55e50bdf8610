#!/usr/bin/env python3
"""What should run(mode="auto") do on an engine with hist=?  Every form that fills T_hist, timed on the same box at 10k / 100k /
1M members (trajectory stored, statistics off: the shape `auto` is used with), 4096 and 1024 bins, fp64 and fp32:
    per_step + bin ring | tiled (K = the auto K, and K = the largest tile) | fused + streamed bin ring
Writes a text table and profiles-style JSON (--json PATH): {dtype: {bins: {members: {form: us_per_step}}}}."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--json", default=None)
ap.add_argument("--members", type=int, nargs="*", default=[10_000, 100_000, 1_000_000])
ap.add_argument("--steps", type=int, default=750)
a = ap.parse_args()
E = emissions.rcp_like_emissions(a.steps, 3)
out = {}
for dt_name, dt in (("f64", torch.float64), ("f32", torch.float32)):
    for nb in (4096, 1024):
        for N in a.members:
            p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=dt)
            eng = EnsembleEngine(p, N, E, dtype=dt, device="cuda:0", hist=(-2.0, 12.0, nb))
            k_auto = max(2, min(eng.auto_k_steps(), eng.tile_steps()))
            forms = [("per_step+bins", dict(mode="per_step")), (f"tiled K={k_auto}", dict(mode="tiled", k_steps=k_auto)),
                     (f"tiled K={eng.tile_steps()}", dict(mode="tiled")), ("fused+bin ring", dict(mode="fused"))]
            res = {}
            for name, kw in forms:
                best = None
                for _ in range(3):
                    eng.reset_state()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    eng.run(**kw)
                    torch.cuda.synchronize()
                    d = (time.perf_counter() - t0) / a.steps * 1e6
                    best = d if best is None else min(best, d)
                res[name] = best
            out.setdefault(dt_name, {}).setdefault(str(nb), {})[str(N)] = {"auto_k_steps": eng.auto_k_steps(), **res}
            print(f"{dt_name} {nb:4d} bins {N:8d} members (auto K = {eng.auto_k_steps():2d}): " +
                  "  ".join(f"{k} {v:7.2f}" for k, v in res.items()) + "  us/step", flush=True)
            eng.close()
            del eng
if a.json:
    with open(a.json, "w") as fh:
        json.dump(out, fh, indent=1)
