#!/usr/bin/env python3
"""Experiment: the per-step kernel on two member halves, each half's launches on its own HIP stream, so that one half's
launch tail / ramp overlaps the other half's kernel.  Same C-ABI entry (sub-range through ld), same results.
    python3 tools/two_stream_per_step.py [members] [f64|f32]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dt = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.float64
dev = torch.device("cuda:0")
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device=dev, dtype=dt)
E = emissions.rcp_like_emissions(750, 3)
eng = EnsembleEngine(p, N, E, dtype=dt, device=dev, chunk_members=0)
fn = getattr(eng.lib, f"fiveeq_run_{eng._sfx}")
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]


def run(parts):
    eng.reset_state()
    torch.cuda.synchronize()
    bounds = [(i * N // parts // 256 * 256, (i + 1) * N // parts // 256 * 256 if i + 1 < parts else N) for i in range(parts)]
    t0 = time.perf_counter()
    if parts == 1:
        assert fn(*eng._run_args(0, 750), eng._stream()) == 0
    else:
        # interleave in time so that both streams always have work queued: steps in blocks of 25 per part
        for t in range(0, 750, 25):
            for i, (lo, hi) in enumerate(bounds):
                assert fn(*eng._run_args(t, t + 25, lo, hi - lo), eng._stream(streams[i])) == 0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 750 * 1e6


ref = None
for parts in (1, 2, 3, 4, 1):
    best = min(run(parts) for _ in range(3))
    T_last = eng.T[-1].clone()
    if ref is None:
        ref = T_last
    print(f"{N} members {'f32' if dt == torch.float32 else 'f64'}: {parts} stream(s) {best:8.2f} us/step   same bits: {torch.equal(T_last, ref)}", flush=True)
