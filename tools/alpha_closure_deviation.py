#!/usr/bin/env python3
"""How far is the closed-form alpha = g0 exp(iIRF/g1) (what the five equations use: cheap, GPU-friendly) from the
alpha Millar et al. 2017 obtain by root-solving  iIRF_100 = sum_i alpha a_i tau_i [1 - exp(-100/(alpha tau_i))] ?

The closed form is the tangent fit at alpha = 1 (tests/test_oracle.py checks the tangency); this script quantifies the
deviation over the alpha range the Latin-hypercube ensemble ACTUALLY visits on the benchmark scenario, per gas.
CPU only (NumPy oracle).   python tools/alpha_closure_deviation.py > profiles/r02/alpha_closure_deviation.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from oracle import fiveeq_oracle as npo  # noqa: E402


def deviation(a, tau, alpha):
    """(alpha_closed_form(iIRF_exact(alpha)) - alpha) / alpha on an array of true alphas."""
    g0, g1 = float(npo.g_0(a, tau)), float(npo.g_1(a, tau))
    iirf = np.array([npo.iirf_exact(x, a, tau) for x in alpha])
    return g0 * np.exp(iirf / g1) / alpha - 1.0


def main(n=4096, verbose=True):
    out = {}
    for kind, names in (("co2", ["CO2"]), ("multigas", ["CO2", "CH4", "N2O"])):
        base = params.default_params(kind)
        p = params.sample_ensemble_shard(base, n)
        E = emissions.rcp_like_emissions(750, len(names))
        alpha = npo.run(E, p, n, keep=("alpha",))["alpha"]          # [750, G, n]
        a = np.atleast_2d(np.asarray(base["a"], dtype=np.float64))
        tau = np.atleast_2d(np.asarray(base["tau"], dtype=np.float64))
        for g, gas in enumerate(names):
            lo, hi = float(alpha[:, g].min()), float(alpha[:, g].max())
            q01, q50, q99 = np.percentile(alpha[:, g], (1, 50, 99))
            grid = np.exp(np.linspace(np.log(lo), np.log(hi), 2001))
            dev = deviation(a[g], tau[g], grid)
            i = int(np.argmax(np.abs(dev)))
            mid = deviation(a[g], tau[g], np.exp(np.linspace(np.log(q01), np.log(q99), 801)))
            out[(kind, gas)] = (lo, hi, float(dev[i]), float(grid[i]), float(np.abs(mid).max()))
            if verbose:
                print(f"{kind:8s} {gas:3s}: alpha visited [{lo:.4f}, {hi:.4f}] (1%/50%/99%: {q01:.4f} / {q50:.4f} / {q99:.4f}); "
                      f"closed form vs root-solve: worst {dev[i]:+.3%} at alpha = {grid[i]:.4f}; "
                      f"within the 1-99 % range <= {np.abs(mid).max():.3%}")
    return out


if __name__ == "__main__":
    print(__doc__.strip().splitlines()[0])
    print(f"{4096} members of the shard-computable Latin hypercube, 750-step RCP-like scenario, oracle alpha per step")
    main()
