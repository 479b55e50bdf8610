#!/usr/bin/env python3
"""A/B of builds of the library (tools: make OUT=/tmp/fiveeq_variants/libfiveeq_X.so EXTRA=-D...): fused fp32 kernel at the
config-5 shard (us/step), ulp of the fp32 exp / expm1 primitives, and fp32-vs-fp64 trajectory differences.
    python3 tools/variant_ab.py lib1.so [lib2.so ...]        ("default" = the in-tree build)"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import _capi, emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

E = emissions.rcp_like_emissions(750, 3)
N_ACC, N_BIG = 200_000, int(os.environ.get("N_BIG", 12_500_000))
rng = np.random.default_rng(7)
x_em1 = -np.concatenate([10.0 ** rng.uniform(-30, 1.9, 1_000_000), rng.uniform(0, 3, 1_000_000)]).astype(np.float32)
x_exp = np.concatenate([rng.uniform(-80, 80, 1_000_000), rng.uniform(-6, 6, 1_000_000)]).astype(np.float32)
p64 = params.sample_ensemble_shard(params.default_params("multigas"), N_ACC, device="cuda:0")
ref = EnsembleEngine(p64, N_ACC, E, device="cuda:0")
ref.run(mode="fused")
torch.cuda.synchronize()
p32 = {k: (v.float() if isinstance(v, torch.Tensor) else v) for k, v in p64.items()}
pbig = params.sample_ensemble_shard(params.default_params("multigas"), N_BIG, device="cuda:0", dtype=torch.float32)
p1m = params.sample_ensemble_shard(params.default_params("multigas"), 1_000_000, device="cuda:0")
first64 = None
for path in sys.argv[1:]:
    lib_path = None if path == "default" else path
    lib = _capi.load(lib_path)
    out = [os.path.basename(path)]
    for op, arr, fn in ((0, x_em1, np.expm1), (1, x_exp, np.exp)):
        xd = torch.from_numpy(arr).cuda()
        yd = torch.empty_like(xd)
        assert lib.fiveeq_math_probe_f32(op, xd.numel(), ctypes.c_void_p(xd.data_ptr()), ctypes.c_void_p(yd.data_ptr()), None) == 0
        torch.cuda.synchronize()
        want = fn(arr.astype(np.float64))
        u = np.abs(yd.cpu().numpy().astype(np.float64) - want) / np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
        u = u[np.isfinite(u) & (np.abs(want) > 1e-37)]
        out.append(f"{'expm1' if op == 0 else 'exp'} ulp max {u.max():.2f} mean {u.mean():.3f}")
    eng = EnsembleEngine(p32, N_ACC, E, dtype=torch.float32, device="cuda:0", lib_path=lib_path)
    eng.run(mode="fused")
    torch.cuda.synchronize()
    for name in ("C", "T"):
        a, b = getattr(eng, name).double(), getattr(ref, name)
        out.append(f"{name} worst rel diff vs fp64 {float(((a - b).abs() / b.abs().clamp_min(1e-3)).max()):.3e}")
    eng.close()
    big = EnsembleEngine(pbig, N_BIG, E, dtype=torch.float32, device="cuda:0", store_trajectory=False, collect_stats=True,
                         lib_path=lib_path)
    best = None
    for _ in range(3):
        big.reset_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        big.run(mode="fused")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out.append(f"fused fp32 {N_BIG} members, stats on: {best / 750 * 1e6:.2f} us/step")
    big.close()
    del big
    for traj in (False, True):
        e64 = EnsembleEngine(p1m, 1_000_000, E, device="cuda:0", store_trajectory=traj, collect_stats=not traj, lib_path=lib_path)
        best = None
        for _ in range(3):
            e64.reset_state()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e64.run(mode="fused")
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out.append(f"fused fp64 1M {'trajectories' if traj else 'stats only'}: {best / 750 * 1e6:.2f} us/step")
        if traj:
            if first64 is None:
                first64 = (e64.C.clone(), e64.T.clone())
            else:
                out.append(f"fp64 bits equal to {os.path.basename(sys.argv[1])}: {torch.equal(e64.C, first64[0]) and torch.equal(e64.T, first64[1])}")
        e64.close()
        del e64
    e64 = EnsembleEngine(p1m, 1_000_000, E, device="cuda:0", lib_path=lib_path)
    best = None
    for _ in range(3):
        e64.reset_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e64.run(mode="per_step")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out.append(f"per-step fp64 1M trajectories: {best / 750 * 1e6:.2f} us/step")
    e64.close()
    del e64
    print(" | ".join(out), flush=True)
