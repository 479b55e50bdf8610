#!/usr/bin/env python3
"""Workload for a kernel trace of the streamed histogram pipeline (fused kernel writing bin indices + histogram pass):
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/ring_pipeline_trace.py [same|side] [S] [members]
prints the wall time per step of the steady-state run; the trace gives the kernels' own durations."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

where = sys.argv[1] if len(sys.argv) > 1 else "same"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N = int(sys.argv[3]) if len(sys.argv) > 3 else 12_500_000
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=torch.float32)
E = emissions.rcp_like_emissions(750, 3)
eng = EnsembleEngine(p, N, E, dtype=torch.float32, store_trajectory=False, collect_stats=True, hist=(-2.0, 12.0, 4096))
eng.hist_ring_steps, eng.hist_pass_stream = S, where
for rep in range(3):
    eng.reset_state()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(mode="fused")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rep {rep}: {dt / 750 * 1e6:.2f} us/step wall ({where} stream, ring 2x{S})", flush=True)
