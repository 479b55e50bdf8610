#!/usr/bin/env python3
"""Same-box A/B of library builds on the in-loop histogram forms at the config-5 shard (12.5M fp32 members, no trajectory,
statistics on): stats-only fused, fused + streamed bin-index ring, per-step + bin ring.
    python3 tools/hist_rule_ab.py default /tmp/fiveeq_variants/lib_X.so ...      (rounds alternate between the builds)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import _capi, emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N = int(os.environ.get("N_BIG", 12_500_000))
STEPS = int(os.environ.get("STEPS", 750))
ROUNDS = int(os.environ.get("ROUNDS", 2))
E = emissions.rcp_like_emissions(STEPS, 3)
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=torch.float32)
CASES = [("fused stats-only", None, dict(mode="fused")),
         ("fused + bin ring 2x64, 4096 bins", 4096, dict(mode="fused")),
         ("per-step stats-only", None, dict(mode="per_step")),
         ("per-step + bin ring, 4096 bins", 4096, dict(mode="per_step"))]
best = {}
for rnd in range(ROUNDS):
    for path in sys.argv[1:]:
        lib_path = None if path == "default" else path
        flags = _capi.build_flags(_capi.load(lib_path)) or "(product build)"
        for name, nb, kw in CASES:
            eng = EnsembleEngine(p, N, E, dtype=torch.float32, device="cuda:0", store_trajectory=False, collect_stats=True,
                                 hist=None if nb is None else (-2.0, 12.0, nb), hist_ring_steps=64, lib_path=lib_path)
            for rep in range(2):
                eng.reset_state()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.run(**kw)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / STEPS * 1e6
                if rep or rnd:
                    key = (path, flags, name)
                    best[key] = min(best.get(key, 1e9), dt)
            eng.close()
            del eng
            torch.cuda.empty_cache()
print(f"{N} members fp32, {STEPS} steps, no trajectory, statistics on; best of {2 * ROUNDS - 1} passes, builds alternating")
for path in sys.argv[1:]:
    rows = [(k, v) for k, v in best.items() if k[0] == path]
    print(f"== {os.path.basename(path)}  {rows[0][0][1]}")
    base = {k[2]: v for k, v in rows}
    for (_, _, name), v in rows:
        ref = base["per-step stats-only"] if name.startswith("per-step") else base["fused stats-only"]
        print(f"   {name:36s} {v:8.2f} us/step  {v / ref - 1:+6.1%}")
