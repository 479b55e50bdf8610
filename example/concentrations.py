#!/usr/bin/env python3
"""BASELINE.json configs[0]: "Single CO2-only run, default params, ~750 yr emissions series via
example/concentrations.py on CPU NumPy (plumbing, no GPU)".

The reference keeps a copy of its one model function in a file of this name
(example/concentrations.py:4-5 of stujen/fiveEqSCM @ v0).  Here the file is a runnable example of
the drop-in: it calls the reference-compatible function on a 750-year series on the CPU, and —
with --gpu, on a machine with an MI355X and the built library — runs the five-equation model the
reference announces for one CO2-only member with the default parameters.

    python example/concentrations.py            # CPU, NumPy
    python example/concentrations.py --gpu      # + single-member CO2-only engine run
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fiveeqscm_amd.concentrations import calculate_hfc_conc  # noqa: E402
from fiveeqscm_amd import emissions, params  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    a = ap.parse_args()

    years = 1765 + np.arange(750)
    time = np.arange(750)
    pulse = np.zeros(750)
    pulse[0] = 10.0
    conc = calculate_hfc_conc(pulse, time, lifetime=1.0)          # the reference's call, verbatim
    print("calculate_hfc_conc: 10-unit pulse in", years[0])
    for t in (0, 1, 2, 3, 10, 100, 745, 746):
        print(f"  t = {t:3d}   conc = {conc[t]:.17g}")

    if a.gpu:
        import torch
        from fiveeqscm_amd.concentrations import run_ensemble
        E = emissions.rcp_like_emissions(750, 1)
        out = run_ensemble(E, params.default_params("co2"), 1)
        C, T = out["C"][:, 0, 0].cpu().numpy(), out["T"][:, 0].cpu().numpy()
        print("five-equation run, CO2 only, default parameters, RCP-like emissions:")
        for t in (0, 100, 200, 282, 400, 749):
            print(f"  {years[t]}   E = {E[t, 0]:6.2f} GtC/yr   C = {C[t]:7.2f} ppm   T = {T[t]:5.3f} K")
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
