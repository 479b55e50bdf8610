#!/usr/bin/env python3
"""The five equations integrated as ORDINARY DIFFERENTIAL EQUATIONS by a general-purpose solver (scipy solve_ivp, DOP853,
rtol = 3e-14) for the 24 golden members: an oracle check that does not share the step algebra.

    python tests/golden/make_fiveeq_ode_reference.py     # build container (scipy); writes fiveeq_ode_reference.json

The product, both oracles and the 50-digit re-evaluation all advance a step with the same closed forms
(R += expm1(-dt/(alpha tau)) (R - a c E alpha tau), S += expm1(-dt/d) (S - q F), and g0 / g1 from their closed expressions).
This script knows none of them.  It knows the model as the papers the reference's README cites (README.md:15) state it:
    dR_i/dt   = a_i c E - R_i / (alpha tau_i)                    (pools; E and alpha constant over a step)
    dcum/dt   = E                                                (cumulative emissions, integrated like everything else)
    dS_j/dt   = (q_j F - S_j) / d_j                              (thermal boxes; F constant over a step)
    alpha     = g0 exp(iIRF / g1),  iIRF = min(r0 + rC G_u + rT T + ra G_a, iirf_max)      evaluated at the START of a step
    F         = sum_g f1 ln(C/C0) + f2 (C - C0) + f3 (sqrt C - sqrt C0)                    evaluated on the pools at its END
and gets g0, g1 from the DEFINITION of the 100-year integrated impulse response by numerical quadrature (scipy quad):
    iIRF100(alpha) = int_0^100 sum_i a_i exp(-t / (alpha tau_i)) dt,   g1 = d iIRF100 / d alpha at alpha = 1
                   = int_0^100 sum_i a_i (t / tau_i) exp(-t / tau_i) dt,   g0 = exp(-iIRF100(1) / g1).
Test infrastructure; the reference itself holds nothing for this path.
"""
import json
import os
import sys

import numpy as np
from scipy.integrate import quad, solve_ivp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import fiveeq_cases as cases  # noqa: E402

RTOL, ATOL = 3e-14, 1e-16


def g_consts(a, tau, H=100.0):
    """(g0, g1) by quadrature of the impulse response (see the module docstring); pools with tau >> H are integrated like
    the others (their integrand is smooth)."""
    irf = lambda t: float(np.sum(a * np.exp(-t / tau)))                      # noqa: E731
    d_irf = lambda t: float(np.sum(a * (t / tau) * np.exp(-t / tau)))        # noqa: E731
    pts = sorted(set(float(v) for v in np.clip(tau, 0.0, H)) - {0.0, H})
    iirf1 = quad(irf, 0.0, H, epsabs=0.0, epsrel=1e-13, points=pts or None, limit=500)[0]
    g1 = quad(d_irf, 0.0, H, epsabs=0.0, epsrel=1e-13, points=pts or None, limit=500)[0]
    return np.exp(-iirf1 / g1), g1


def run(p, E, n_members, dt=1.0, n_steps=None):
    """C [n_steps, G, N], T [n_steps, N] of the first n_steps steps, all members integrated as one ODE system per step."""
    a_all = np.atleast_2d(np.asarray(p["a"], dtype=np.float64))
    tau_all = np.atleast_2d(np.asarray(p["tau"], dtype=np.float64))
    G, N = a_all.shape[0], int(n_members)
    n_steps = E.shape[0] if n_steps is None else int(n_steps)
    pools = [int(np.nonzero(a_all[g])[0][-1]) + 1 for g in range(G)]
    a = [a_all[g, :pools[g]] for g in range(G)]
    tau = [tau_all[g, :pools[g]] for g in range(G)]
    g0g1 = [g_consts(a[g], tau[g]) for g in range(G)]
    col = lambda k: np.asarray(p[k], dtype=np.float64).reshape(G)            # noqa: E731
    ra, C0, c, f = col("ra"), col("PI_conc"), col("emis2conc"), np.asarray(p["f"], dtype=np.float64).reshape(G, 3)
    r0, rC, rT, q = (np.asarray(p[k], dtype=np.float64) for k in ("r0", "rC", "rT", "q"))
    d = np.asarray(p["d"], dtype=np.float64)
    iirf_max = float(p["iirf_max"])
    R = [np.zeros((pools[g], N)) for g in range(G)]
    cum = np.zeros((G, N))
    S = np.zeros((2, N))
    C_out, T_out = np.zeros((n_steps, G, N)), np.zeros((n_steps, N))
    sizes = [pools[g] * N for g in range(G)]
    for t in range(n_steps):
        T_old = S[0] + S[1]
        alpha = []
        for g in range(G):
            G_a = R[g].sum(0) / c[g]
            G_u = cum[g] - G_a
            iirf = np.minimum(r0[g] + rC[g] * G_u + rT[g] * T_old + ra[g] * G_a, iirf_max)
            alpha.append(g0g1[g][0] * np.exp(iirf / g0g1[g][1]))

        def pools_rhs(_, y, t=t, alpha=alpha):
            out, o = [], 0
            for g in range(G):
                Rg = y[o:o + sizes[g]].reshape(pools[g], N)
                out.append((a[g][:, None] * (c[g] * E[t, g]) - Rg / (alpha[g][None, :] * tau[g][:, None])).ravel())
                o += sizes[g]
            out.append(np.repeat(E[t, :G], N))                                # d cum / dt = E
            return np.concatenate(out)

        y0 = np.concatenate([R[g].ravel() for g in range(G)] + [cum.ravel()])
        sol = solve_ivp(pools_rhs, (0.0, dt), y0, method="DOP853", rtol=RTOL, atol=ATOL)
        assert sol.success, sol.message
        y, o = sol.y[:, -1], 0
        for g in range(G):
            R[g] = y[o:o + sizes[g]].reshape(pools[g], N)
            o += sizes[g]
        cum = y[o:].reshape(G, N)
        F = np.zeros(N)
        for g in range(G):
            Cg = C0[g] + R[g].sum(0)
            C_out[t, g] = Cg
            F += f[g, 0] * np.log(Cg / C0[g]) + f[g, 1] * (Cg - C0[g]) + f[g, 2] * (np.sqrt(Cg) - np.sqrt(C0[g]))
        sol = solve_ivp(lambda _, s, F=F: ((q * F[None, :] - s.reshape(2, N)) / d[:, None]).ravel(), (0.0, dt), S.ravel(),
                        method="DOP853", rtol=RTOL, atol=ATOL)
        assert sol.success, sol.message
        S = sol.y[:, -1].reshape(2, N)
        T_out[t] = S[0] + S[1]
    return C_out, T_out


def main():
    import scipy
    doc = {"generator": "tests/golden/make_fiveeq_ode_reference.py", "scipy": scipy.__version__, "numpy": np.__version__,
           "method": "DOP853", "rtol": RTOL, "atol": ATOL, "steps": cases.STEPS, "cases": {}}
    for kind in ("co2", "multigas"):
        p, N = cases.members(kind)
        C, T = run(p, cases.scenario(kind), N)
        doc["cases"][kind] = {"C": [[[float(v).hex() for v in C[t, g]] for g in range(C.shape[1])] for t in cases.STEPS],
                              "T": [[float(v).hex() for v in T[t]] for t in cases.STEPS]}
        print(kind, "done", flush=True)
    path = os.path.join(HERE, "fiveeq_ode_reference.json")
    with open(path, "w") as fh:
        json.dump(doc, fh, separators=(",", ":"))
        fh.write("\n")
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
