#!/usr/bin/env python3
"""A 50-digit re-evaluation of the five-equation recurrence (mpmath) for a handful of golden members: bounds the
rounding error the fp64 oracle accumulates over 750 steps, so that "the kernels agree with the oracle to 1e-10" is a
statement about ACCURACY, not only about agreement between two fp64 programs.

    python tests/golden/make_fiveeq_mp_reference.py     # build container (mpmath); writes fiveeq_mp_reference.json

Inputs are exactly the fp64 numbers the oracle receives (parameters, emissions); everything derived from them —
g0, g1, cumulative emissions, expm1(-dt/d) — is recomputed in 50 digits.  This is a third, independent statement
of the model step (test infrastructure, like everything under oracle/ and tests/); the reference has none.
"""
import json
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import fiveeq_cases as cases  # noqa: E402

mp.mp.dps = 50
MEMBERS = list(range(24))                 # ALL golden members: 16 of the Latin hypercube and the 8 corners of the box
M = mp.mpf


def g_consts(a, tau, H=100):
    g1 = mp.fsum(ai * ti * (1 - (1 + H / ti) * mp.exp(-H / ti)) for ai, ti in zip(a, tau))
    iirf_unit = mp.fsum(ai * ti * (1 - mp.exp(-H / ti)) for ai, ti in zip(a, tau))
    return mp.exp(-iirf_unit / g1), g1


def run_member(p, E, m, dt=1):
    a_all = np.atleast_2d(np.asarray(p["a"], dtype=np.float64))
    tau_all = np.atleast_2d(np.asarray(p["tau"], dtype=np.float64))
    G = a_all.shape[0]
    gases = []
    for g in range(G):
        P = int(np.nonzero(a_all[g])[0][-1]) + 1
        a = [M(float(v)) for v in a_all[g, :P]]
        tau = [M(float(v)) for v in tau_all[g, :P]]
        g0, g1 = g_consts(a, tau)
        f = [M(float(v)) for v in np.asarray(p["f"], dtype=np.float64).reshape(G, 3)[g]]
        gases.append(dict(a=a, tau=tau, g0=g0, g1=g1, ra=M(float(np.asarray(p["ra"], dtype=np.float64).reshape(G)[g])),
                          C0=M(float(np.asarray(p["PI_conc"], dtype=np.float64).reshape(G)[g])),
                          c=M(float(np.asarray(p["emis2conc"], dtype=np.float64).reshape(G)[g])), f=f,
                          r0=M(float(p["r0"][g, m])), rC=M(float(p["rC"][g, m])), rT=M(float(p["rT"][g, m])),
                          R=[M(0)] * P, cum=M(0)))
    d = [M(float(v)) for v in p["d"]]
    q = [M(float(p["q"][j, m])) for j in range(2)]
    em1_d = [mp.expm1(-M(dt) / dj) for dj in d]
    iirf_max = M(float(p["iirf_max"]))
    S = [M(0), M(0)]
    Cs, Ts = [], []
    for t in range(E.shape[0]):
        T_old = S[0] + S[1]
        F = M(0)
        row = []
        for g, gs in enumerate(gases):
            Eg = M(float(E[t, g]))
            G_a = mp.fsum(gs["R"]) / gs["c"]
            G_u = gs["cum"] - G_a
            iirf = min(gs["r0"] + gs["rC"] * G_u + gs["rT"] * T_old + gs["ra"] * G_a, iirf_max)
            alpha = gs["g0"] * mp.exp(iirf / gs["g1"])
            gs["R"] = [Ri + mp.expm1(-M(dt) / (alpha * ti)) * (Ri - ai * gs["c"] * Eg * alpha * ti)
                       for Ri, ai, ti in zip(gs["R"], gs["a"], gs["tau"])]
            C = gs["C0"] + mp.fsum(gs["R"])
            F += gs["f"][0] * mp.log(C / gs["C0"]) + gs["f"][1] * (C - gs["C0"]) + gs["f"][2] * (mp.sqrt(C) - mp.sqrt(gs["C0"]))
            gs["cum"] += Eg * dt
            row.append(C)
        S = [Sj + e * (Sj - qj * F) for Sj, e, qj in zip(S, em1_d, q)]
        Cs.append(row)
        Ts.append(S[0] + S[1])
    return Cs, Ts


def _one(arg):
    kind, m = arg
    mp.mp.dps = 50
    p, _ = cases.members(kind)
    Cs, Ts = run_member(p, cases.scenario(kind), m)
    return ([[mp.nstr(Cs[t][g], 30) for g in range(len(Cs[t]))] for t in cases.STEPS],
            [mp.nstr(Ts[t], 30) for t in cases.STEPS])


def main():
    doc = {"generator": "tests/golden/make_fiveeq_mp_reference.py", "mpmath": mp.__version__, "digits": mp.mp.dps,
           "members": MEMBERS, "steps": cases.STEPS, "cases": {}}
    import multiprocessing
    with multiprocessing.Pool(min(6, os.cpu_count() or 1)) as pool:       # members are independent: a few at a time
        for kind in ("co2", "multigas"):
            rec = {"C": [], "T": []}
            for C_rows, T_row in pool.map(_one, [(kind, m) for m in MEMBERS]):
                rec["C"].append(C_rows)
                rec["T"].append(T_row)
            doc["cases"][kind] = rec
    path = os.path.join(HERE, "fiveeq_mp_reference.json")
    with open(path, "w") as fh:
        json.dump(doc, fh, separators=(",", ":"))
        fh.write("\n")
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
