"""Host-side parameter preparation for the five-equation ensemble engine.

Function names follow the only trace of the intended split in the reference
(stujen/fiveEqSCM @ v0), the commented names at .coveragerc:15-18:
`g_1`, `g_0`, `alpha_val`, `k_q`.  The reference ships no parameter set
("Appropriate tunings and parameter sets will be made available in due
course", README.md:10), so the sets below are THIS build's choice; their
provenance is written beside each number.

The "parameter dict" (the engine's and the oracle's common input):

    shared (same for every member)
      a          [G,4]  pool fractions (active pools = leading non-zero entries)
      tau        [G,4]  pool time-scales, yr
      ra         [G]    iIRF sensitivity to the gas's own burden G_a
      PI_conc    [G]    pre-industrial concentration C0
      emis2conc  [G]    concentration units per emission unit
      f          [G,3]  forcing coefficients (log, linear, sqrt)
      iirf_max   float  clip on iIRF_100
      d          [2]    thermal-box time-scales, yr
    per member (shape [G,N] / [2,N]) or shared (shape [G] / [2])
      r0, rC, rT [G]    iIRF_100 = r0 + rC*G_u + rT*T + ra*G_a
      q          [2]    thermal-box coefficients, K / (W m^-2)
"""
import math

import numpy as np

from . import _capi

LHS_SEED = 20261003


# ------------------------------------------------------------------------------------
# closed-form alpha constants
# ------------------------------------------------------------------------------------
def _one_minus_1px_exp(x):
    """1 - (1+x) e^-x without cancellation (x > 0, scalar)."""
    if x < 0.05:
        term, total, m = x * x, 0.0, 2
        sign = 1.0
        # sum_{m>=2} (-1)^m (m-1)/m! x^m
        fact = 2.0
        while True:
            add = sign * (m - 1) / fact * term
            total += add
            if abs(add) < 1e-18 * abs(total) or m > 24:
                return total
            m += 1
            fact *= m
            term *= x
            sign = -sign
    return 1.0 - (1.0 + x) * math.exp(-x)


def _active(a_row, tau_row):
    a_row = [float(v) for v in np.asarray(a_row, dtype=np.float64).ravel()]
    tau_row = [float(v) for v in np.asarray(tau_row, dtype=np.float64).ravel()]
    n = 0
    for i, v in enumerate(a_row):
        if v != 0.0:
            n = i + 1
    if n == 0 or any(v == 0.0 for v in a_row[:n]):
        raise ValueError("active pools must be the leading, non-zero entries of a")
    return a_row[:n], tau_row[:n]


def g_1(a, tau, H=100.0):
    """g1 = sum_i a_i tau_i [1 - (1 + H/tau_i) exp(-H/tau_i)]."""
    a, tau = _active(a, tau)
    return math.fsum(ai * ti * _one_minus_1px_exp(H / ti) for ai, ti in zip(a, tau))


def g_0(a, tau, H=100.0):
    """g0 = exp(-sum_i a_i tau_i [1 - exp(-H/tau_i)] / g1)."""
    aa, tt = _active(a, tau)
    iirf_unit = math.fsum(ai * ti * (-math.expm1(-H / ti)) for ai, ti in zip(aa, tt))
    return math.exp(-iirf_unit / g_1(a, tau, H))


def alpha_val(G_u, G_a, T, r0, rC, rT, ra, g0, g1, iirf_max):
    """alpha = g0 * exp(min(r0 + rC*G_u + rT*T + ra*G_a, iirf_max) / g1)  (host helper, NumPy)."""
    iirf = np.minimum(r0 + rC * G_u + rT * T + ra * G_a, iirf_max)
    return g0 * np.exp(iirf / g1)


def forcing(C, C0, f):
    """f1 ln(C/C0) + f2 (C - C0) + f3 (sqrt C - sqrt C0) for C > 0 (host helper, scalar)."""
    return f[0] * math.log(C / C0) + f[1] * (C - C0) + f[2] * (math.sqrt(C) - math.sqrt(C0))


def k_q(TCR, ECS, d, F2x):
    """(TCR, ECS) -> q [2, ...]:  ECS = F2x (q1 + q2),  TCR = F2x (q1 k1 + q2 k2),
    k_j = 1 - (d_j/70)(1 - exp(-70/d_j))."""
    d = np.asarray(d, dtype=np.float64)
    k = 1.0 - (d / 70.0) * (-np.expm1(-70.0 / d))
    TCR = np.asarray(TCR, dtype=np.float64)
    ECS = np.asarray(ECS, dtype=np.float64)
    den = F2x * (k[0] - k[1])
    return np.stack([(TCR - ECS * k[1]) / den, (ECS * k[0] - TCR) / den], axis=0)


# ------------------------------------------------------------------------------------
# default parameter sets (shared values; r0/rC/rT/q here are the ensemble CENTRES)
# ------------------------------------------------------------------------------------
def default_params(kind="co2"):
    """`co2`: CO2 only, Millar et al. 2017 values (the paper README.md:15 of the reference
    cites), as recalled in SURVEY.md section 8c.  `multigas`: CO2 + CH4 + N2O, the CO2 row
    as above with a log+sqrt forcing, CH4/N2O single-pool rows with values of the size
    published for FaIR v2.0 (Leach et al. 2021) — recalled, not verifiable offline;
    this build's choice."""
    co2 = dict(
        a=[0.2173, 0.2240, 0.2824, 0.2763],
        tau=[1.0e6, 394.4, 36.54, 4.304],
        r0=32.4, rC=0.019, rT=4.165, ra=0.0,
        PI_conc=278.0,
        emis2conc=1.0 / 2.123,           # ppm per GtC
    )
    if kind == "co2":
        F2x = 3.74
        return {
            "a": [co2["a"]], "tau": [co2["tau"]],
            "r0": [co2["r0"]], "rC": [co2["rC"]], "rT": [co2["rT"]], "ra": [co2["ra"]],
            "PI_conc": [co2["PI_conc"]], "emis2conc": [co2["emis2conc"]],
            "f": [[F2x / math.log(2.0), 0.0, 0.0]],
            "iirf_max": 97.0,
            "d": [239.0, 4.1],
            "q": [0.33, 0.41],
        }
    if kind == "multigas":
        # mass of atmosphere 5.1352e18 kg, mean molar mass 28.97 g/mol -> 1.7726e11 mol per ppb
        mol_per_ppb = 5.1352e18 / 28.97e-3 * 1e-9
        return {
            "a": [co2["a"], [1.0, 0.0, 0.0, 0.0], [1.0, 0.0, 0.0, 0.0]],
            "tau": [co2["tau"], [9.15, 1.0, 1.0, 1.0], [116.0, 1.0, 1.0, 1.0]],
            "r0": [co2["r0"], 9.08, 67.8],
            "rC": [co2["rC"], 0.0, 0.0],
            "rT": [co2["rT"], -0.287, 0.0],
            "ra": [0.0, 3.2e-4, 0.0],
            "PI_conc": [278.0, 720.0, 270.0],         # ppm, ppb, ppb
            "emis2conc": [co2["emis2conc"],
                          1.0 / (mol_per_ppb * 16.04e-3 / 1e9),   # ppb per Mt CH4
                          1.0 / (mol_per_ppb * 28.01e-3 / 1e9)],  # ppb per Mt N2O-N2
            "f": [[4.57, 0.0, 0.086], [0.0, 0.0, 0.038], [0.0, 0.0, 0.106]],
            "iirf_max": 97.0,
            "d": [239.0, 4.1],
            "q": [0.33, 0.41],
        }
    raise ValueError(f"unknown parameter set {kind!r}")


def n_gas_of(params):
    return int(np.atleast_2d(np.asarray(params["a"], dtype=np.float64)).shape[0])


def pools_of(params):
    a = np.atleast_2d(np.asarray(params["a"], dtype=np.float64))
    tau = np.atleast_2d(np.asarray(params["tau"], dtype=np.float64))
    return [len(_active(a[g], tau[g])[0]) for g in range(a.shape[0])]


def forcing_2x(params):
    """Forcing of a CO2 doubling (gas 0) under the set's own forcing coefficients."""
    C0 = float(np.asarray(params["PI_conc"], dtype=np.float64).reshape(-1)[0])
    f = np.asarray(params["f"], dtype=np.float64).reshape(-1, 3)[0]
    return forcing(2.0 * C0, C0, f)


# ------------------------------------------------------------------------------------
# Latin-hypercube ensemble draws
# ------------------------------------------------------------------------------------
def latin_hypercube(n, n_dim, seed=LHS_SEED):
    """[n_dim, n] stratified uniforms in (0,1): one random permutation + jitter per dimension."""
    rng = np.random.default_rng(seed)
    u = np.empty((n_dim, n), dtype=np.float64)
    for k in range(n_dim):
        u[k] = (rng.permutation(n) + rng.random(n)) / n
    return u


def sample_ensemble(base, n_members, seed=LHS_SEED):
    """Perturb r0 (x0.8..1.2), rC, rT (x0.5..1.5) per gas and TCR in [1,2.5] K, ECS in [1.5,4.5] K
    (swapped where ECS < TCR, then ECS >= 1.1 TCR), q from k_q.  Returns a new parameter dict
    whose r0/rC/rT are [G,N] and q is [2,N]; everything else is shared with `base`."""
    G = n_gas_of(base)
    N = int(n_members)
    u = latin_hypercube(N, 3 * G + 2, seed)
    out = dict(base)
    for j, (name, lo, hi) in enumerate((("r0", 0.8, 1.2), ("rC", 0.5, 1.5), ("rT", 0.5, 1.5))):
        centre = np.asarray(base[name], dtype=np.float64).reshape(G)
        out[name] = centre[:, None] * (lo + (hi - lo) * u[j * G:(j + 1) * G])
    tcr = 1.0 + 1.5 * u[3 * G]
    ecs = 1.5 + 3.0 * u[3 * G + 1]
    lo_, hi_ = np.minimum(tcr, ecs), np.maximum(tcr, ecs)
    swap = ecs < tcr
    tcr = np.where(swap, lo_, tcr)
    ecs = np.where(swap, hi_, ecs)
    ecs = np.maximum(ecs, 1.1 * tcr)
    out["q"] = k_q(tcr, ecs, base["d"], forcing_2x(base))
    out["TCR"] = tcr
    out["ECS"] = ecs
    return out


# ------------------------------------------------------------------------------------
# Shard-computable Latin hypercube: the design of the multi-GPU runs.
# `latin_hypercube` above permutes [0, N) with a generator, so every rank would have to draw all N
# members to find its own; here the permutation is a KEYED BIJECTION evaluated member by member
# (cycle-walked 4-round Feistel network, splitmix64 round function), so a rank computes exactly
# its shard — on its GPU through fiveeq_lhs_rows_f64, or on the host with the NumPy twin below,
# bit for bit the same numbers — and the design does not depend on the world size.
# ------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix64(z):
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _lhs_half_bits(n_total):
    bits = 2
    while bits < 62 and (1 << bits) < n_total:
        bits += 2
    return bits // 2


def _lhs_dim_key(seed, dim):
    with np.errstate(over="ignore"):
        return _mix64(np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.uint64(dim + 1))


def lhs_permute(m, n_total, key):
    """pi(m): keyed bijection of [0, n_total) — include/fiveeq.h, fiveeq_lhs_rows_f64."""
    half = np.uint64(_lhs_half_bits(n_total))
    mask = np.uint64((1 << int(half)) - 1)
    x = np.array(m, dtype=np.uint64, copy=True)
    todo = np.ones(x.shape, dtype=bool)
    with np.errstate(over="ignore"):
        while todo.any():
            v = x[todo]
            left, right = v >> half, v & mask
            for rnd in range(4):
                f = _mix64(right ^ (key + np.uint64(0xD1342543DE82EF95) * np.uint64(rnd + 1))) & mask
                left, right = right, left ^ f
            v = (left << half) | right
            x[todo] = v
            todo[todo] = v >= np.uint64(n_total)
    return x


def lhs_rows(n_total, dims, lo=0, hi=None, seed=LHS_SEED):
    """[len(dims), hi-lo] fp64: u_d(m) = (pi_d(m) + jitter_d(m)) / n_total for members lo <= m < hi of a
    Latin hypercube over n_total members; one member per stratum in every dimension."""
    hi = n_total if hi is None else hi
    if not 1 <= n_total <= _capi.LHS_MAX_TOTAL:          # stratum (<= 28 bits) + mid-cell jitter (25 bits): an exact fp64 sum
        raise ValueError(f"n_total={n_total} outside 1..2^28")
    if not 0 <= lo <= hi <= n_total:
        raise ValueError(f"members [{lo}, {hi}) outside [0, {n_total})")
    m = np.arange(lo, hi, dtype=np.uint64)
    out = np.empty((len(dims), hi - lo), dtype=np.float64)
    with np.errstate(over="ignore"):
        for k, d in enumerate(dims):
            key = _lhs_dim_key(seed, int(d))
            stratum = lhs_permute(m, n_total, key)
            jbits = _mix64(m ^ (key * np.uint64(0xFF51AFD7ED558CCD) + np.uint64(0xC4CEB9FE1A85EC53))) >> np.uint64(40)
            jitter = (jbits.astype(np.float64) + 0.5) * 2.0 ** -24
            out[k] = (stratum.astype(np.float64) + jitter) / float(n_total)
    return out


def lhs_rows_device(n_total, dims, lo, hi, device, seed=LHS_SEED):
    """The same rows computed on `device` by the HIP kernel (dims must be consecutive)."""
    import ctypes

    import torch
    dims = list(dims)
    if dims != list(range(dims[0], dims[0] + len(dims))):
        raise ValueError("dims must be consecutive")
    lib = _capi.load()
    dev = torch.device(device)
    out = torch.empty((len(dims), hi - lo), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.fiveeq_lhs_rows_f64(int(seed), int(n_total), int(lo), int(hi - lo), dims[0], len(dims), hi - lo,
                                     ctypes.c_void_p(out.data_ptr()),
                                     ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _capi.check(lib, rc)
    return out


def sample_ensemble_shard(base, n_total, lo=0, hi=None, seed=LHS_SEED, device=None, dtype=None):
    """Members [lo, hi) of the n_total-member design: the perturbation rule of `sample_ensemble` (r0 x0.8..1.2,
    rC, rT x0.5..1.5 per gas; TCR in [1,2.5] K, ECS in [1.5,4.5] K, swapped where ECS < TCR, then
    ECS >= 1.1 TCR; q from k_q) on the shard-computable hypercube above.  O(hi - lo) work and memory.
    device=None: NumPy rows on the host; a torch device: rows are produced on that GPU (no PCIe, no
    host copy) and every per-member entry is a device tensor.  Both give bit-identical numbers, and
    concatenating the shards of any partition of [0, n_total) gives the single-shard result."""
    G = n_gas_of(base)
    hi = n_total if hi is None else hi
    dims = list(range(3 * G + 2))
    if device is None:
        u = lhs_rows(n_total, dims, lo, hi, seed)
        xp_min, xp_max, xp_where, xp_stack = np.minimum, np.maximum, np.where, np.stack
    else:
        import torch
        u = lhs_rows_device(n_total, dims, lo, hi, device, seed)
        xp_min, xp_max, xp_where, xp_stack = torch.minimum, torch.maximum, torch.where, torch.stack
    out = dict(base)
    for j, (name, a, b) in enumerate((("r0", 0.8, 1.2), ("rC", 0.5, 1.5), ("rT", 0.5, 1.5))):
        centre = np.asarray(base[name], dtype=np.float64).reshape(G)
        # one rounding per operation on both back ends: (b - a) * u, + a, * centre
        out[name] = xp_stack([((b - a) * u[j * G + g] + a) * float(centre[g]) for g in range(G)])
    tcr = 1.5 * u[3 * G] + 1.0
    ecs = 3.0 * u[3 * G + 1] + 1.5
    lo_, hi_ = xp_min(tcr, ecs), xp_max(tcr, ecs)
    swap = ecs < tcr
    tcr = xp_where(swap, lo_, tcr)
    ecs = xp_max(xp_where(swap, hi_, ecs), 1.1 * tcr)
    d = np.asarray(base["d"], dtype=np.float64)
    k = 1.0 - (d / 70.0) * (-np.expm1(-70.0 / d))
    # multiply by the reciprocal on both back ends (torch turns a division by a scalar into that on the GPU
    # anyway): one rounding per operation, identical on host and device
    inv_den = 1.0 / float(forcing_2x(base) * (k[0] - k[1]))
    out["q"] = xp_stack([(tcr - ecs * float(k[1])) * inv_den, (ecs * float(k[0]) - tcr) * inv_den])
    out["TCR"], out["ECS"] = tcr, ecs
    if dtype is not None and device is not None:
        for name in ("r0", "rC", "rT", "q"):
            out[name] = out[name].to(dtype)
    return out


# ------------------------------------------------------------------------------------
# pack the shared part into the C-ABI struct
# ------------------------------------------------------------------------------------
def make_model(params, dt=1.0):
    """Parameter dict -> ctypes `fiveeq_model` (include/fiveeq.h)."""
    a = np.atleast_2d(np.asarray(params["a"], dtype=np.float64))
    tau = np.atleast_2d(np.asarray(params["tau"], dtype=np.float64))
    G = a.shape[0]
    if not 1 <= G <= _capi.MAX_GAS:
        raise ValueError(f"n_gas={G} outside 1..{_capi.MAX_GAS}")
    if a.shape != tau.shape or a.shape[1] > _capi.MAX_POOLS:
        raise ValueError(f"a {a.shape} / tau {tau.shape}: want [G,<=4]")
    ra = np.asarray(params["ra"], dtype=np.float64).reshape(G)
    C0 = np.asarray(params["PI_conc"], dtype=np.float64).reshape(G)
    c = np.asarray(params["emis2conc"], dtype=np.float64).reshape(G)
    f = np.asarray(params["f"], dtype=np.float64).reshape(G, 3)
    d = np.asarray(params["d"], dtype=np.float64).reshape(_capi.N_BOX)
    m = _capi.Model()
    m.n_gas = G
    m.dt = float(dt)
    m.iirf_max = float(params["iirf_max"])
    for j in range(_capi.N_BOX):
        m.d[j] = float(d[j])
    for g in range(G):
        aa, tt = _active(a[g], tau[g])
        gs = m.gas[g]
        gs.n_pools = len(aa)
        for i, (ai, ti) in enumerate(zip(aa, tt)):
            gs.a[i] = ai
            gs.tau[i] = ti
        gs.g0 = g_0(a[g], tau[g])
        gs.g1 = g_1(a[g], tau[g])
        gs.ra = float(ra[g])
        gs.C0 = float(C0[g])
        gs.emis2conc = float(c[g])
        for k in range(3):
            gs.f[k] = float(f[g, k])
    return m
