#!/usr/bin/env python3
"""Small ensembles (BASELINE configs[1]: 10k CO2-only fp64 members): the small-ensemble kernel (one member per quad of
lanes, lanes 4 / the same kernel unspread, lanes 1) against the fused kernel and the K-step form; every form is first
checked torch.equal against the per-step path.  us per model step, whole 750-step scenario passes, HIP events.
    python3 tools/small_ensemble_ab.py [members ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

dtype = torch.float32 if "--f32" in sys.argv else torch.float64
kind, G = ("multigas", 3) if "--multigas" in sys.argv else ("co2", 1)
stats = "--stats" in sys.argv                       # per-64-member statistics records on, in every form
sizes = [int(v) for v in sys.argv[1:] if not v.startswith("--")] or [10_000, 20_000, 40_000, 80_000, 160_000, 320_000]
E = emissions.rcp_like_emissions(750, G)
print(f"# {torch.cuda.get_device_name(0)}, dtype {dtype}, {kind}, 750 steps, trajectories stored{', statistics on' if stats else ''}; us per step (median of 7 passes)")
print(f"# {'members':>8} {'fused':>8} {'fused1':>8} {'ksteps':>8} {'small1':>8} {'small4':>8}   small4 member-steps/s   equal")
for N in sizes:
    p = params.sample_ensemble_shard(params.default_params(kind), N, device="cuda:0", dtype=dtype)
    eng = EnsembleEngine(p, N, E, dtype=dtype, collect_stats=stats)
    eng.run(mode="per_step")
    torch.cuda.synchronize()
    ref = [eng.C.clone(), eng.T.clone(), eng.R.clone(), eng.S.clone()]
    out, same = {}, True
    forms = [("fused", "fused", {}), ("fused1", "fused", {"span": None}), ("ksteps", "ksteps", {}), ("small1", "small", {"lanes": 1})]
    forms += [("small4", "small", {"lanes": eng.small_widest})] if eng.small_widest in (4, 8) and not stats else []      # (column "small4": the quad / octet form)
    out["small4"] = float("nan")
    for name, mode, kw in forms:
        eng.small_lanes = kw.get("lanes", "auto")
        eng.fused_span = kw.get("span", "auto")
        ts = []
        for rep in range(8):
            eng.reset_state()
            eng.C.zero_(), eng.T.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.run(mode=mode)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 750)
        out[name] = float(np.median(ts[1:]))
        ok = all(torch.equal(a, b) for a, b in zip(ref, [eng.C, eng.T, eng.R, eng.S]))
        same = same and ok
        if not ok:
            print(f"   {name}: NOT bit-identical to the per-step path", [float((a - b).abs().max()) for a, b in zip(ref, [eng.C, eng.T, eng.R, eng.S])])
    print(f"  {N:8d} {out['fused']:8.3f} {out['fused1']:8.3f} {out['ksteps']:8.3f} {out['small1']:8.3f} {out['small4']:8.3f}   "
          f"{N / out['small4'] * 1e6:.3e}          {same}")
    eng.close()
    del eng, p
