#!/usr/bin/env python3
"""Generate golden vectors for `calculate_hfc_conc` by importing the reference.

Run ONLY in the build container, where /root/reference is mounted:

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 \
        python tests/golden/make_hfc_golden.py

Writes tests/golden/hfc_conc_golden.json.  The JSON holds inputs and expected
outputs only (fp64 as C99 hex-float strings, so the fixture is bit-exact and
text-diffable).  The reference module itself never travels: nothing under
tests/ or on the GPU box imports it.

Reference function: U_FaIR/concentrations.py:4-5 (duplicate at
example/concentrations.py:4-5); reference test: tests/unit/test_hfcs.py:5-13.
"""
import hashlib
import json
import os
import sys

import numpy as np

import U_FaIR.concentrations as _ref_module  # the reference, unmodified
from U_FaIR.concentrations import calculate_hfc_conc

# this repo ships a `U_FaIR/concentrations.py` of its own (a re-export of the drop-in): make sure the vectors
# below really come from the reference's file, never from that shim
assert os.path.realpath(_ref_module.__file__).startswith("/root/reference/"), _ref_module.__file__


def hexlist(x):
    return [float(v).hex() for v in np.asarray(x, dtype=np.float64).ravel()]


def case(name, emissions, time, lifetime, note=""):
    out = calculate_hfc_conc(emissions, time, lifetime=lifetime)
    out_arr = np.asarray(out, dtype=np.float64)
    return {
        "name": name,
        "note": note,
        "emissions": np.asarray(emissions).tolist(),
        "emissions_dtype": str(np.asarray(emissions).dtype),
        "time": np.asarray(time).tolist(),
        "time_dtype": str(np.asarray(time).dtype),
        "lifetime": lifetime,
        "out_shape": list(out_arr.shape),
        "out_is_scalar": bool(np.ndim(out) == 0),
        "out_dtype": str(out_arr.dtype),
        "out_hex": hexlist(out_arr),
        "out_sha256": hashlib.sha256(out_arr.tobytes()).hexdigest(),
    }


def main():
    cases = []
    # (1) the reference's own known-answer test, tests/unit/test_hfcs.py:5-13
    cases.append(case("ref_unit_test", np.array([10, 0, 0, 0]), np.array([0, 1, 2, 3]), 1.0,
                      "tests/unit/test_hfcs.py:6-10"))
    # (2) lifetime is accepted but ignored (U_FaIR/concentrations.py:5 never reads it)
    cases.append(case("lifetime_ignored", np.array([10, 0, 0, 0]), np.array([0, 1, 2, 3]), 123.0))
    # (3) only emissions[0] is read
    cases.append(case("only_first_emission", np.array([10, 7, -3, 99]), np.array([0, 1, 2, 3]), 1.0))
    # (4) BASELINE config-1 shape: 750-step series, int64 and float64 time
    e750 = np.zeros(750)
    e750[0] = 10
    cases.append(case("config1_int_time", e750, np.arange(750), 1.0, "subnormals from ~t=709, zero at t>=746"))
    cases.append(case("config1_float_time", e750, np.arange(750, dtype=np.float64), 1.0))
    # (5) fractional / negative / non-monotone time: time is an absolute coordinate
    cases.append(case("fractional_time", np.array([2.5]), np.array([0.0, 0.5, 1.25, -1.0, 3.0, 0.5]), 4.0))
    # (6) scalar time -> scalar out
    cases.append(case("scalar_time", np.array([10.0, 1.0]), np.float64(2.0), 1.0))
    # (7) 2-D time -> 2-D out
    cases.append(case("time_2d", np.array([3.0]), np.arange(6, dtype=np.float64).reshape(2, 3), 1.0))
    # (8) 2-D emissions (3,4) with time (4,): row 0 broadcast against time
    cases.append(case("emissions_2d", np.arange(12, dtype=np.float64).reshape(3, 4) + 1.0,
                      np.array([0.0, 1.0, 2.0, 3.0]), 1.0))
    # (9) list emissions accepted
    cases.append(case("list_emissions", [4, 0, 0], np.array([0, 2, 4]), 1.0))

    # behavioural pin: empty emissions -> IndexError
    try:
        calculate_hfc_conc(np.array([]), np.array([0, 1]), lifetime=1.0)
        empty = "no error"
    except Exception as exc:  # noqa: BLE001 - we record exactly what the reference raises
        empty = type(exc).__name__

    doc = {
        "generator": "tests/golden/make_hfc_golden.py",
        "reference": "stujen/fiveEqSCM @ v0: U_FaIR/concentrations.py:4-5",
        "numpy": np.__version__,
        "python": sys.version.split()[0],
        "empty_emissions_raises": empty,
        "cases": cases,
    }
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hfc_conc_golden.json")
    with open(path, "w") as fh:
        json.dump(doc, fh, indent=1)
    print("wrote", path, "cases:", len(cases), "empty ->", empty)


if __name__ == "__main__":
    main()
