"""Drop-in for the reference module `U_FaIR/concentrations.py` (stujen/fiveEqSCM @ v0),
plus the ensemble entry points that the reference announces but does not have.

`calculate_hfc_conc` keeps the reference's name, parameter names and behaviour exactly
(U_FaIR/concentrations.py:4-5; duplicate example/concentrations.py:4-5), quirks
included: `lifetime` is accepted and ignored, only `emissions[0]`
is read, `time` is an absolute coordinate, empty `emissions` raises IndexError.  It is
NumPy-in / NumPy-out and CPU-only, like the reference (BASELINE.json configs[0]:
"plumbing, no GPU").
"""
import ctypes

import numpy as np


def calculate_hfc_conc(emissions, time, lifetime):
    """emissions[0] * exp(-time)  — behaviour of U_FaIR/concentrations.py:5."""
    return emissions[0] * np.exp(-time)


def calculate_hfc_conc_ensemble(e0, time, device=None):
    """Ensemble form on the GPU: out[k, m] = e0[m] * exp(-time[k]) through the C ABI
    (`fiveeq_hfc_conc_f64`).  e0: [N] first-year emissions per member; time: [n_time].
    Returns a device tensor [n_time, N] (fp64).  GPU only; raises without the HIP library."""
    import torch

    from . import _capi

    lib = _capi.load()
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: calculate_hfc_conc_ensemble has no CPU fallback")
    dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    e0_t = torch.as_tensor(np.asarray(e0, dtype=np.float64)).reshape(-1).to(dev).contiguous()
    if e0_t.numel() == 0:
        raise IndexError("emissions is empty")      # the reference raises IndexError here too
    tm = torch.as_tensor(np.asarray(time, dtype=np.float64)).reshape(-1).to(dev).contiguous()
    N, K = e0_t.numel(), tm.numel()
    out = torch.empty((K, N), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        rc = lib.fiveeq_hfc_conc_f64(N, N, K, ctypes.c_void_p(e0_t.data_ptr()), ctypes.c_void_p(tm.data_ptr()),
                                     ctypes.c_void_p(out.data_ptr()),
                                     ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _capi.check(lib, rc)
    return out


def run_ensemble(emissions, params, n_members, **kwargs):
    """Five-equation ensemble run on the GPU; see fiveeqscm_amd.engine.run_ensemble."""
    from .engine import run_ensemble as _run

    return _run(emissions, params, n_members, **kwargs)
