#!/usr/bin/env python3
"""Golden trajectories of the five-equation path, generated from THE BUILD'S OWN ORACLE (oracle/fiveeq_oracle.py).

The reference (stujen/fiveEqSCM @ v0) has no implementation, test or vector for this path ("parity unpinned",
SURVEY.md section 8c), so these vectors do not pin parity with the reference.  What they pin is the oracle ITSELF: with
them committed, a change to oracle/fiveeq_oracle.py (or .c) and to the kernels together can no longer pass every test
silently — the fixture would have to be regenerated, and that shows in the diff.

    python tests/golden/make_fiveeq_golden.py          # from the repo root; writes tests/golden/fiveeq_golden.json

Content: for 'co2' and 'multigas', 24 members (16 Latin-hypercube members + 8 corners of the perturbation box,
tests/golden/fiveeq_cases.py) x 35 selected steps of the 750-step scenario: C [G], T, alpha [G] as C99 hex floats, the
final pools and boxes; plus one concentration-driven (inverse) case (diagnosed emissions).  fp64, bit-exact text.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import fiveeq_cases as cases  # noqa: E402
from oracle import fiveeq_oracle as npo  # noqa: E402


def hx(x):
    return [float(v).hex() for v in np.asarray(x, dtype=np.float64).ravel()]


def main():
    doc = {"generator": "tests/golden/make_fiveeq_golden.py", "source": "oracle/fiveeq_oracle.py (the build's own oracle)",
           "numpy": np.__version__, "steps": cases.STEPS, "cases": {}}
    for kind in ("co2", "multigas"):
        p, N = cases.members(kind)
        E = cases.scenario(kind)
        out = npo.run(E, p, N, keep=("C", "T", "alpha"))
        rec = {"n_members": N, "n_steps": int(E.shape[0]),
               "r0": hx(p["r0"]), "rC": hx(p["rC"]), "rT": hx(p["rT"]), "q": hx(p["q"]),
               "C": hx(out["C"][cases.STEPS]), "T": hx(out["T"][cases.STEPS]), "alpha": hx(out["alpha"][cases.STEPS]),
               "R_final": hx(np.concatenate(out["R"], axis=0)), "S_final": hx(out["S"]),
               "sha256_C_T_all_steps": hashlib.sha256(out["C"].tobytes() + out["T"].tobytes()).hexdigest()}
        doc["cases"][kind] = rec
    # inverse mode: drive the CO2-only members with the concentration pathway of member 0's forward run
    p, N = cases.members("co2")
    E = cases.scenario("co2")
    fwd = npo.run(E, p, N, keep=("C",))
    conc = fwd["C"][:, :, 0]                                   # [750, 1]
    inv = npo.run_inverse(conc, p, N)
    doc["cases"]["co2_inverse"] = {"n_members": N, "target_member": 0, "E": hx(inv["E"][cases.STEPS]),
                                   "T": hx(inv["T"][cases.STEPS]), "cumE_final": hx(inv["cumE"])}
    # the same for the multi-gas set: all three gases driven by member 5's concentration pathways
    p, N = cases.members("multigas")
    E = cases.scenario("multigas")
    conc = npo.run(E, p, N, keep=("C",))["C"][:, :, 5]         # [750, 3]
    inv = npo.run_inverse(conc, p, N)
    doc["cases"]["multigas_inverse"] = {"n_members": N, "target_member": 5, "E": hx(inv["E"][cases.STEPS]),
                                        "T": hx(inv["T"][cases.STEPS]), "cumE_final": hx(inv["cumE"])}
    # a run with a time step of half a year and an external forcing ramp (F_ext), CO2 only
    p, N = cases.members("co2")
    E2 = np.repeat(cases.scenario("co2"), 2, axis=0)[:600]     # 600 half-year steps
    Fx = 0.002 * np.arange(600)
    sub = npo.run(E2, p, N, F_ext=Fx, dt=0.5, keep=("C", "T"))
    steps2 = [s for s in cases.STEPS if s < 600]
    doc["cases"]["co2_halfyear_fext"] = {"n_members": N, "n_steps": 600, "dt": 0.5, "steps": steps2,
                                         "C": hx(sub["C"][steps2]), "T": hx(sub["T"][steps2])}
    path = os.path.join(HERE, "fiveeq_golden.json")
    with open(path, "w") as fh:
        json.dump(doc, fh, separators=(",", ":"))
        fh.write("\n")
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
