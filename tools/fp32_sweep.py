#!/usr/bin/env python3
"""fp64-vs-fp32 tolerance sweep (BASELINE configs[4]: "100M-member fp32 ensemble ... with
fp64-vs-fp32 tolerance sweep reported"; SURVEY.md section 8d: on a 10^6-member subset).

Runs the same Latin-hypercube members through the fp64 and the fp32 kernels and reports, per output
year, the distribution over members of |x32 - x64| / |x64| for C (per gas) and T, plus the effect on
the ensemble statistics that are the product of such a run (mean, 5/50/95th percentiles of T)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=1_000_000)
    ap.add_argument("--mode", default="per_step")
    a = ap.parse_args()
    N = a.members
    p = params.sample_ensemble(params.default_params("multigas"), N)
    E = emissions.rcp_like_emissions(750, 3)
    years = [0, 99, 249, 399, 499, 749]
    e64 = EnsembleEngine(p, N, E, dtype=torch.float64, device="cuda:0", output_steps=years)
    e64.run(mode="fused")
    torch.cuda.synchronize()
    names = ["CO2", "CH4", "N2O"]
    # the default fp32 arithmetic (every launch form gives these bits) and the COMPENSATED fp32 form of the time-fused kernel
    # (round 6: a compensation word per pool and box in registers + the forcing from the excess C - C0; include/fiveeq.h)
    for label, kw, mode in (("fp32 (default arithmetic)", {}, a.mode), ("fp32 COMPENSATED (time-fused kernel)", {"compensated": True}, "fused")):
        e32 = EnsembleEngine(p, N, E, dtype=torch.float32, device="cuda:0", output_steps=years, **kw)
        e32.run(mode=mode)
        torch.cuda.synchronize()
        print(f"{label} vs fp64, {N} members, 750 steps, CO2+CH4+N2O, mode={mode}")
        print(f"{'year':>5s} {'qty':>4s} {'median rel':>11s} {'p99 rel':>11s} {'max rel':>11s} {'max abs':>11s}")
        worst = 0.0
        for k, t in enumerate(years):
            rows = [(names[g], e32.C[k, g].double(), e64.C[k, g]) for g in range(3)] + [("T", e32.T[k].double(), e64.T[k])]
            for name, x32, x64 in rows:
                err = (x32 - x64).abs()
                rel = err / x64.abs().clamp_min(1e-30)
                q = torch.quantile(rel[:: max(1, N // 1_000_000)], torch.tensor([0.5, 0.99], dtype=torch.float64, device=rel.device))
                print(f"{t:5d} {name:>4s} {q[0].item():11.3e} {q[1].item():11.3e} {rel.max().item():11.3e} {err.max().item():11.3e}")
                if t > 50:
                    worst = max(worst, rel.max().item())
        print(f"worst relative difference after year 50: {worst:.3e}")
        print("effect on ensemble statistics of T (fp32 - fp64):")
        for k, t in enumerate(years):
            if t < 200:
                continue
            s32, s64 = torch.sort(e32.T[k].double()).values, torch.sort(e64.T[k]).values
            idx = [int(f * (N - 1)) for f in (0.05, 0.5, 0.95)]
            d = [(s32[i] - s64[i]).item() for i in idx]
            print(f"  year {t}: mean {(e32.T[k].double().mean() - e64.T[k].mean()).item():+.3e} K, "
                  f"p05 {d[0]:+.3e}, p50 {d[1]:+.3e}, p95 {d[2]:+.3e} K")
        e32.close()
        print()


if __name__ == "__main__":
    main()
