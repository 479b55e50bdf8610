#!/usr/bin/env python3
"""Same box, same process: round 2's library (built from git f943ad9 into build_variants/libfiveeq_r02.so) against the
current one on the kernels round 3 changed — fused fp32 at the config-5 shard (12.5M members, no trajectory, statistics
on / off), per-step fp32 at 1M, and the fp64 kernels as the control (unchanged code: the ratio is the box's noise).
    python3 tools/r02_vs_r03.py"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import _capi, emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

old = ctypes.CDLL(os.path.join(ROOT, "build_variants", "libfiveeq_r02.so"))
for name in ("fiveeq_run_fused_f32", "fiveeq_run_fused_f64", "fiveeq_run_f32", "fiveeq_run_f64"):
    fn = getattr(old, name)
    fn.restype, fn.argtypes = _capi.SIGNATURES[name]
E = emissions.rcp_like_emissions(750, 3)


def timed(eng, fn, reps=3):
    best = None
    for _ in range(reps + 1):
        eng.reset_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        assert fn(*eng._run_args(0, eng.n_steps), eng._stream()) == 0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best / eng.n_steps * 1e6


for label, N, dt, kw, entry in (
        ("fused fp32 12.5M, stats on, nothing stored", 12_500_000, torch.float32, dict(store_trajectory=False, collect_stats=True), "fiveeq_run_fused_f32"),
        ("fused fp32 12.5M, stats off, nothing stored", 12_500_000, torch.float32, dict(store_trajectory=False), "fiveeq_run_fused_f32"),
        ("per-step fp32 1M, trajectories stored", 1_000_000, torch.float32, dict(), "fiveeq_run_f32"),
        ("fused fp64 1M, stats on, nothing stored (control)", 1_000_000, torch.float64, dict(store_trajectory=False, collect_stats=True), "fiveeq_run_fused_f64"),
        ("per-step fp64 1M, trajectories stored (control)", 1_000_000, torch.float64, dict(), "fiveeq_run_f64")):
    p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=dt)
    eng = EnsembleEngine(p, N, E, dtype=dt, device="cuda:0", **kw)
    eng._wave_stats()
    t_old = timed(eng, getattr(old, entry))
    t_new = timed(eng, getattr(eng.lib, entry))
    print(f"{label:<52} round 2 {t_old:8.2f} us/step   round 3 {t_new:8.2f} us/step   {t_new / t_old - 1:+.1%}", flush=True)
    eng.close()
    del eng, p
