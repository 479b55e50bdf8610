#!/usr/bin/env python3
"""Which of the streams torch hands out run beside the default stream (HIP maps streams onto a few hardware queues in creation
order; two streams on one queue serialise), and what a two-stream per-step run costs on one that does not — the reason the
engine probes its side streams (fiveeqscm_amd/tuning.py, concurrent_side_streams).
    python3 tools/side_stream_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import _capi, emissions, params, tuning  # noqa: E402
from fiveeqscm_amd import engine as engine_mod  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

lib = _capi.load()
main = torch.cuda.current_stream()
N, c = 4_000_000, 1_228_800
E = emissions.rcp_like_emissions(750, 3)[200:280]
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0")


def per_step_us(side):
    """us per step of a chunk-major two-stream per-step run whose side stream is `side` (None: the engine's own choice)."""
    picked = engine_mod.concurrent_side_streams
    if side is not None:
        engine_mod.concurrent_side_streams = lambda lib_, main_, count: [side][:count]
    try:
        eng = EnsembleEngine(p, N, E, device="cuda:0", chunk_members=c, per_step_streams=2)
        best = None
        for _ in range(3):
            eng.reset_state()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run(0, 80, mode="per_step")
            torch.cuda.synchronize()
            d = (time.perf_counter() - t0) / 80 * 1e6
            best = d if best is None else min(best, d)
        eng.close()
        return best
    finally:
        engine_mod.concurrent_side_streams = picked


print(f"# {torch.cuda.get_device_name(0)}; {N} fp64 members, chunk-major ({c} per chunk), two streams; the caller's stream is the default stream")
print("# stream torch hands out | runs beside the default stream (probe) | us per step with it as the side stream")
for i in range(12):
    s = torch.cuda.Stream()
    print(f"  {i:2d}  {s.cuda_stream:#014x}   {str(tuning.streams_concurrent(lib, main, s)):5s}   {per_step_us(s):7.1f}", flush=True)
print("# the engine's own choice (probed once per process and caller's stream), eight engines in a row:")
print("  " + "  ".join(f"{per_step_us(None):6.1f}" for _ in range(8)))
