"""Synthetic RCP-like emissions and the shared per-step "drive" table.

The reference (stujen/fiveEqSCM @ v0) ships no emissions data (SURVEY.md
section 2, negative inventory); the deterministic series below are the ones
frozen in SURVEY.md section 8d.  They are shared by every ensemble member.
"""
import hashlib

import numpy as np

from ._capi import DRIVE_STRIDE, MAX_GAS

N_STEPS_DEFAULT = 750   # years 1765..2514, dt = 1 yr


def _sigma(x):
    return 1.0 / (1.0 + np.exp(-x))


def rcp_like_emissions(n_steps=N_STEPS_DEFAULT, n_gas=3):
    """[n_steps, n_gas] fp64.  Columns: CO2 (GtC/yr), CH4 (Mt CH4/yr), N2O (Mt N2O-N2/yr).

    CO2: 12 s((t-230)/25) (1 - 1.1 s((t-330)/20)), clipped at >= -1  (peak ~9.7 GtC/yr near
    step 282, -1 GtC/yr from ~step 400: an overshoot scenario);
    CH4: 300 s((t-200)/40) (1 - 0.5 s((t-330)/30));   N2O: 10 s((t-220)/50);   s = logistic.
    """
    if not 1 <= n_gas <= MAX_GAS:
        raise ValueError(f"n_gas={n_gas} outside 1..{MAX_GAS}")
    t = np.arange(n_steps, dtype=np.float64)
    co2 = np.maximum(12.0 * _sigma((t - 230.0) / 25.0) * (1.0 - 1.1 * _sigma((t - 330.0) / 20.0)), -1.0)
    ch4 = 300.0 * _sigma((t - 200.0) / 40.0) * (1.0 - 0.5 * _sigma((t - 330.0) / 30.0))
    n2o = 10.0 * _sigma((t - 220.0) / 50.0)
    return np.stack([co2, ch4, n2o][:n_gas], axis=1)


def emissions_sha256(E):
    return hashlib.sha256(np.ascontiguousarray(E, dtype=np.float64).tobytes()).hexdigest()


def make_drive(emissions, F_ext=None, dt=1.0, output_steps=None, concentration_driven=False):
    """emissions [n_steps, G] (or [n_steps]) -> drive [n_steps, 8] fp64 (include/fiveeq.h):
    cols 0..2 E_g, cols 3..5 cumulative emissions BEFORE the step, col 6 F_ext, col 7 the output
    row the step's C/T are stored at (-1: not stored).  output_steps=None stores every step at
    row t; otherwise the listed steps are stored at rows 0, 1, ... in increasing step order.
    concentration_driven=True: `emissions` holds target concentrations (end of step) for the
    inverse mode; cols 3..5 stay zero."""
    E = np.asarray(emissions, dtype=np.float64)
    if E.ndim == 1:
        E = E[:, None]
    if E.ndim != 2 or not 1 <= E.shape[1] <= MAX_GAS or E.shape[0] < 1:
        raise ValueError(f"emissions shape {E.shape}: want [n_steps>=1, 1..{MAX_GAS}]")
    if not np.all(np.isfinite(E)):
        raise ValueError("emissions contain non-finite values")
    n_steps, G = E.shape
    drive = np.zeros((n_steps, DRIVE_STRIDE), dtype=np.float64)
    drive[:, :G] = E
    if not concentration_driven:          # inverse mode: cumulative emissions are per-member state
        drive[1:, 3:3 + G] = np.cumsum(E * dt, axis=0)[:-1]
    if F_ext is not None:
        F_ext = np.asarray(F_ext, dtype=np.float64).reshape(-1)
        if F_ext.shape[0] != n_steps:
            raise ValueError(f"F_ext has {F_ext.shape[0]} steps, emissions {n_steps}")
        drive[:, 6] = F_ext
    if output_steps is None:
        drive[:, 7] = np.arange(n_steps)
    else:
        steps = np.unique(np.asarray(output_steps, dtype=np.int64).reshape(-1))
        if steps.size and (steps[0] < 0 or steps[-1] >= n_steps):
            raise ValueError(f"output_steps outside [0, {n_steps})")
        drive[:, 7] = -1.0
        drive[steps, 7] = np.arange(steps.size)
    return drive
