"""ctypes wrapper of oracle/libfiveeq_oracle.so (the plain-C restatement, fiveeq_oracle.c).

TEST INFRASTRUCTURE ONLY (see the header of fiveeq_oracle.py): used by tests/, smoke() and
bench.py's cpu_baseline leg.  Builds its model struct from the parameter dict with the
ORACLE's own g_0 / g_1, independently of fiveeqscm_amd.params.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import fiveeq_oracle as npo

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfiveeq_oracle.so")


class _Gas(ctypes.Structure):
    _fields_ = [("a", ctypes.c_double * 4), ("tau", ctypes.c_double * 4), ("g0", ctypes.c_double),
                ("g1", ctypes.c_double), ("ra", ctypes.c_double), ("C0", ctypes.c_double),
                ("emis2conc", ctypes.c_double), ("f", ctypes.c_double * 3), ("n_pools", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class _Model(ctypes.Structure):
    _fields_ = [("gas", _Gas * 3), ("d", ctypes.c_double * 2), ("iirf_max", ctypes.c_double),
                ("dt", ctypes.c_double), ("n_gas", ctypes.c_int32), ("reserved", ctypes.c_int32)]


_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.oracle_run.restype = ctypes.c_int
        _lib.oracle_hfc_conc.restype = ctypes.c_int
        _lib.oracle_max_threads.restype = ctypes.c_int
    return _lib


def make_model(params, dt=1.0):
    a = np.atleast_2d(np.asarray(params["a"], dtype=np.float64))
    tau = np.atleast_2d(np.asarray(params["tau"], dtype=np.float64))
    G = a.shape[0]
    m = _Model()
    m.n_gas = G
    m.dt = dt
    m.iirf_max = float(params["iirf_max"])
    d = np.asarray(params["d"], dtype=np.float64).reshape(2)
    m.d[0], m.d[1] = float(d[0]), float(d[1])
    ra = np.asarray(params["ra"], dtype=np.float64).reshape(G)
    C0 = np.asarray(params["PI_conc"], dtype=np.float64).reshape(G)
    c = np.asarray(params["emis2conc"], dtype=np.float64).reshape(G)
    f = np.asarray(params["f"], dtype=np.float64).reshape(G, 3)
    for g in range(G):
        P = npo.n_pools_of(a[g])
        gs = m.gas[g]
        gs.n_pools = P
        for i in range(P):
            gs.a[i] = float(a[g, i])
            gs.tau[i] = float(tau[g, i])
        gs.g0 = float(npo.g_0(a[g], tau[g]))
        gs.g1 = float(npo.g_1(a[g], tau[g]))
        gs.ra, gs.C0, gs.emis2conc = float(ra[g]), float(C0[g]), float(c[g])
        for k in range(3):
            gs.f[k] = float(f[g, k])
    return m


def _p(x):
    return x.ctypes.data_as(ctypes.c_void_p) if x is not None else ctypes.c_void_p(0)


def run(emissions, params, n_members, F_ext=None, dt=1.0, R0=None, S0=None, keep=("C", "T"), n_threads=1,
        t_begin=0, t_end=None):
    """Same contract as fiveeq_oracle.run(): returns dict C [n_steps,G,N], T [n_steps,N], R [SP,N], S [2,N]."""
    lib = load()
    N = int(n_members)
    drive = np.ascontiguousarray(npo.make_drive(emissions, F_ext, dt))
    n_steps = drive.shape[0]
    mdl = make_model(params, dt)
    G = mdl.n_gas
    SP = sum(mdl.gas[g].n_pools for g in range(G))
    r = np.ascontiguousarray(np.concatenate(
        [np.stack([npo._member_rows(params[k], G, N)[g] for k in ("r0", "rC", "rT")]) for g in range(G)], axis=0))
    q = np.ascontiguousarray(npo._member_rows(params["q"], 2, N))
    R = np.zeros((SP, N)) if R0 is None else np.ascontiguousarray(np.asarray(R0, dtype=np.float64).reshape(SP, N)).copy()
    S = np.zeros((2, N)) if S0 is None else np.ascontiguousarray(np.asarray(S0, dtype=np.float64).reshape(2, N)).copy()
    C = np.empty((n_steps, G, N)) if "C" in keep else None
    T = np.empty((n_steps, N)) if "T" in keep else None
    t_end = n_steps if t_end is None else t_end
    rc = lib.oracle_run(ctypes.byref(mdl), ctypes.c_int64(N), ctypes.c_int64(N), _p(drive), ctypes.c_int32(n_steps),
                        ctypes.c_int32(t_begin), ctypes.c_int32(t_end), _p(r), _p(q), _p(R), _p(S), _p(C), _p(T),
                        ctypes.c_int32(n_threads))
    if rc != 0:
        raise RuntimeError(f"oracle_run returned {rc}")
    return {"C": C, "T": T, "R": R, "S": S}


def hfc_conc(e0, time):
    lib = load()
    e0 = np.ascontiguousarray(e0, dtype=np.float64).reshape(-1)
    time = np.ascontiguousarray(time, dtype=np.float64).reshape(-1)
    out = np.empty((time.size, e0.size))
    rc = lib.oracle_hfc_conc(ctypes.c_int64(e0.size), ctypes.c_int64(e0.size), ctypes.c_int32(time.size),
                             _p(e0), _p(time), _p(out))
    if rc != 0:
        raise RuntimeError(f"oracle_hfc_conc returned {rc}")
    return out


def max_threads():
    return int(load().oracle_max_threads())
