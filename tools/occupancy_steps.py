#!/usr/bin/env python3
"""How many 256-thread workgroups of the fused packed-fp32 kernel does a CU really hold?  us/step against the number of
workgroups per CU (members = 256 CUs x k x 512), statistics on, nothing stored: the time steps up when k passes the limit.
    python3 tools/occupancy_steps.py [lib.so]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "default" else None
E = emissions.rcp_like_emissions(750, 3)
for k in (1, 2, 3, 4, 5, 6, 8, 16):
    N = 256 * k * 512
    p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=torch.float32)
    eng = EnsembleEngine(p, N, E, dtype=torch.float32, store_trajectory=False, collect_stats=True, lib_path=lib)
    best = None
    for _ in range(4):
        eng.reset_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(mode="fused")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print(f"{k:3d} workgroups of 256 per CU ({N:9d} members): fused {best / 750 * 1e6:7.3f} us/step  = {best / 750 * 1e6 / k:6.3f} per workgroup-round", flush=True)
    del eng, p
