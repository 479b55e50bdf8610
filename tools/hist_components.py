#!/usr/bin/env python3
"""Components of the streamed-histogram pipeline at the config-5 shard: fused with/without T rows, hist_rows alone."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
dt = torch.float32 if (len(sys.argv) < 3 or sys.argv[2] == "f32") else torch.float64
steps = 128
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=dt)
E = emissions.rcp_like_emissions(750, 3)[300:300 + steps]


def timed(fn, reps=4):
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        best = d if best is None else min(best, d)
    return best / steps * 1e6


for stats in (True, False):
    a = EnsembleEngine(p, N, E, dtype=dt, store_trajectory=False, collect_stats=stats)
    print(f"stats={int(stats)} fused, nothing stored      {timed(lambda: (a.reset_state(), a.run(mode='fused'))):8.2f} us/step")
    del a
    b = EnsembleEngine(p, N, E, dtype=dt, store_concentrations=False, collect_stats=stats)
    print(f"stats={int(stats)} fused, T of every step     {timed(lambda: (b.reset_state(), b.run(mode='fused'))):8.2f} us/step")
    if stats:
        for nb in (4096, 1024, 64):
            print(f"   hist_rows alone, {nb:4d} bins        {timed(lambda: b.T_histogram(-2.0, 12.0, nb)):8.2f} us/row")
        x = b.T
        print(f"   torch sum over rows (read-only pass) {timed(lambda: x.sum()):8.2f} us/row")
    del b
