"""Constant-lifetime minor gases (HFCs, PFCs, SF6, ...) as a shared forcing series.

The reference's one function models exactly such a species: a reservoir that decays with a fixed
lifetime, `emissions[0] * exp(-time)` (U_FaIR/concentrations.py:4-5).  In the five-equation
framework these gases are the special case of one pool with no state dependence (rC = rT = ra = 0,
alpha = 1): linear, and — because their emissions and parameters are shared by every ensemble member
— identical for all members.  They therefore need no per-member state on the GPU at all: their
concentrations are advanced once on the host with the same exact-step formula the kernels use,
    R <- R + expm1(-dt/tau) * (R - tau * c * E),     C = C0 + R,
and their summed forcing  sum_k eff_k * (C_k - C0_k)  enters the ensemble run as `F_ext`.
"""
import numpy as np


def step_minor_gases(emissions, lifetime, emis2conc, rad_eff, C0=0.0, R0=0.0, dt=1.0):
    """emissions [n_steps, K] (emission units / yr); lifetime [K] yr; emis2conc [K] concentration units per
    emission unit; rad_eff [K] W m^-2 per concentration unit; C0, R0 scalars or [K].
    Returns (conc [n_steps, K] at the END of each step, forcing [n_steps] = sum_k rad_eff_k (C_k - C0_k))."""
    E = np.asarray(emissions, dtype=np.float64)
    if E.ndim == 1:
        E = E[:, None]
    n_steps, K = E.shape
    tau = np.broadcast_to(np.asarray(lifetime, dtype=np.float64), (K,))
    c = np.broadcast_to(np.asarray(emis2conc, dtype=np.float64), (K,))
    eff = np.broadcast_to(np.asarray(rad_eff, dtype=np.float64), (K,))
    C0 = np.broadcast_to(np.asarray(C0, dtype=np.float64), (K,))
    if np.any(tau <= 0) or not np.all(np.isfinite(E)):
        raise ValueError("lifetimes must be > 0 and emissions finite")
    em1 = np.expm1(-dt / tau)
    R = np.array(np.broadcast_to(np.asarray(R0, dtype=np.float64), (K,)), dtype=np.float64)
    conc = np.empty((n_steps, K))
    for t in range(n_steps):
        R = R + em1 * (R - tau * c * E[t])
        conc[t] = C0 + R
    return conc, (conc - C0) @ eff
