#!/usr/bin/env python3
"""Cost of the in-loop histograms on one ensemble, no trajectory stored, statistics on: the
stats-only fused run, the fused kernel + the streamed ring of 2-byte bin indices (ring lengths, pass on the side / same
stream), the per-step kernel with and without its ring strip.  With --small also BASELINE configs[1] (10k CO2-only members)
through every launch form.

    python tools/hist_forms_bench.py [--members 12500000] [--dtype f32] [--steps 750] [--reps 3] [--small]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402


def timed(eng, reps, **kw):
    best = None
    for _ in range(reps + 1):                      # first pass warms clocks / code objects
        eng.reset_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(**kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=12_500_000)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--steps", type=int, default=750)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--kind", default="multigas")
    ap.add_argument("--small", action="store_true", help="also run config 2 through every launch form")
    a = ap.parse_args()
    dt = torch.float32 if a.dtype == "f32" else torch.float64
    G = 3 if a.kind == "multigas" else 1
    N = a.members
    p = params.sample_ensemble_shard(params.default_params(a.kind), N, device="cuda:0", dtype=dt)
    E = emissions.rcp_like_emissions(a.steps, G)
    base = EnsembleEngine(p, N, E, dtype=dt, store_trajectory=False, collect_stats=True)
    t_fused = timed(base, a.reps, mode="fused")
    print(f"{torch.cuda.get_device_name(0)}: {N} members {a.dtype} {a.steps} steps, no trajectory, stats on")
    print(f"  fused (stats only)            {t_fused / a.steps * 1e6:9.2f} us/step  {N * a.steps / t_fused:.3e} member-steps/s")
    t_k32 = timed(base, a.reps, mode="ksteps", k_steps=32)
    print(f"  fused kernel relaunched every 32 steps (ksteps)  {t_k32 / a.steps * 1e6:9.2f} us/step  {t_k32 / t_fused - 1:+.1%} vs fused")
    del base
    for nb in (4096, 1024):
        eng = EnsembleEngine(p, N, E, dtype=dt, store_trajectory=False, collect_stats=True, hist=(-2.0, 12.0, nb))
        for S in (16, 32, 64, 128):
            for where in (("side", "same") if S in (64, 128) else ("side",)):
                eng.hist_ring_steps, eng._bins, eng.hist_pass_stream = S, None, where
                t = timed(eng, a.reps, mode="fused")
                print(f"  fused + streamed hist {nb:4d} bins, bin-index ring 2x{S:3d} steps, pass on the {where} stream "
                      f"{t / a.steps * 1e6:9.2f} us/step  {t / t_fused - 1:+.1%} vs fused   ring {2 * S * N * 2 / 1e9:.2f} GB")
        eng.hist_pass_stream = "side"
        assert eng.T_hist.sum(1).min().item() == N
        del eng
    ps = EnsembleEngine(p, N, E, dtype=dt, store_trajectory=False, collect_stats=True)
    t_ps = timed(ps, 1, mode="per_step")
    del ps
    ph = EnsembleEngine(p, N, E, dtype=dt, store_trajectory=False, collect_stats=True, hist=(-2.0, 12.0, 4096))
    t_ph = timed(ph, 1, mode="per_step")
    assert ph.T_hist.sum(1).min().item() == N
    print(f"  per-step kernel (stats only)  {t_ps / a.steps * 1e6:9.2f} us/step")
    print(f"  per-step kernel + bin-index strip of {ph.hist_ring_steps} steps, 4096 bins {t_ph / a.steps * 1e6:9.2f} us/step  "
          f"{t_ph / t_ps - 1:+.1%} vs per-step")
    if a.small:
        N2 = 10_000
        p2 = params.sample_ensemble_shard(params.default_params("co2"), N2)
        E2 = emissions.rcp_like_emissions(750, 1)
        eng = EnsembleEngine(p2, N2, E2)
        print(f"config 2: {N2} members CO2-only fp64, trajectories stored; auto -> {eng.resolve_mode('auto')}, small_form {eng.small_form()}")
        for mode, k in (("per_step", None), ("graph", None), ("ksteps", 8), ("ksteps", 32), ("fused", None), ("small", 1),
                        ("small", 4), ("auto", None)):
            if mode == "small":
                eng.small_lanes, k_ = k, None
            else:
                eng.small_lanes, k_ = "auto", k
            t = timed(eng, 5, mode=mode, k_steps=k_)
            print(f"  {mode:8s} {'lanes' if mode == 'small' else 'K'}={str(k):4s} {t / 750 * 1e6:7.3f} us/step  {N2 * 750 / t:.3e} member-steps/s")


if __name__ == "__main__":
    main()
