#!/usr/bin/env python3
"""Components of the bin-index-ring pipeline at the config-5 shard: the fused kernel with and without the in-kernel bin
indices, the bin pass and the T pass alone (us per step / per row)."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N, S, n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000, 64, 128
dev = torch.device("cuda:0")
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device=dev, dtype=torch.float32)
E = emissions.rcp_like_emissions(750, 3)[300:300 + n_steps]
eng = EnsembleEngine(p, N, E, dtype=torch.float32, device=dev, store_trajectory=False, collect_stats=True,
                     hist=(-2.0, 12.0, 4096), hist_ring_steps=S)
lib, vp = eng.lib, ctypes.c_void_p
eng._wave_stats()
bins = torch.empty((S, N), dtype=torch.int16, device=dev)
Tring = torch.empty((S, N), dtype=torch.float32, device=dev)
ring_drive = eng._hist_ring(slots=1)["drive"]


def timed(fn, reps=3):
    best = None
    for _ in range(reps):
        eng.reset_state()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        d = time.perf_counter() - t0
        best = d if best is None else min(best, d)
    return best


def chunks(call):
    for t in range(0, n_steps, S):
        assert call(t, min(n_steps, t + S)) == 0


st = eng._stream()
plain = timed(lambda: chunks(lambda t, t1: lib.fiveeq_run_fused_f32(*eng._run_args(t, t1), st)))
withb = timed(lambda: chunks(lambda t, t1: lib.fiveeq_run_fused_bins_f32(*eng._run_args(t, t1), -2.0, 12.0, 4096, vp(bins.data_ptr()), S, st)))
a = eng._run_args(0, S)
withT = timed(lambda: chunks(lambda t, t1: lib.fiveeq_run_fused_f32(ctypes.byref(eng.model), N, N, vp(ring_drive.data_ptr()), n_steps, t, t1, vp(eng.r.data_ptr()), vp(eng.q.data_ptr()), vp(eng.R.data_ptr()), vp(eng.S.data_ptr()), vp(0), vp(Tring.data_ptr()), S, vp(eng.T_stats.data_ptr()), st)))
print(f"{N} members fp32, chunks of {S} steps, statistics on, {n_steps} steps from scenario step 300")
print(f"  fused, nothing stored                         {plain / n_steps * 1e6:8.2f} us/step")
print(f"  fused + bin index of every step (2 B)         {withb / n_steps * 1e6:8.2f} us/step  {withb / plain - 1:+.1%}")
print(f"  fused + T of every step (4 B)                 {withT / n_steps * 1e6:8.2f} us/step  {withT / plain - 1:+.1%}")
hist = torch.zeros((S, 4096), dtype=torch.int64, device=dev)
tb = timed(lambda: lib.fiveeq_hist_bins(S, N, N, vp(bins.data_ptr()), 4096, vp(hist.data_ptr()), st))
tT = timed(lambda: lib.fiveeq_hist_rows_f32(S, N, N, vp(Tring.data_ptr()), -2.0, 12.0, 4096, vp(hist.data_ptr()), st))
print(f"  pass over {S} rows of bin indices                {tb / S * 1e6:8.2f} us/row   {2 * N / (tb / S) / 1e12:.2f} TB/s")
print(f"  pass over {S} rows of T                          {tT / S * 1e6:8.2f} us/row   {4 * N / (tT / S) / 1e12:.2f} TB/s")
