#!/usr/bin/env python3
"""Workload for the FETCH_SIZE / WRITE_SIZE passes of the FUSED kernel: calibration copies (known bytes), then whole
750-step fused launches of the bench workload (1M members, fp64, trajectories stored): per launch the algorithmic bytes are
A_fused x N x 750 = 32.288 B x 1e6 x 750 = 24.2 GB."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N, COPY = 1_000_000, 1 << 27
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0")
eng = EnsembleEngine(p, N, emissions.rcp_like_emissions(750, 3), device="cuda:0")
src = torch.empty(COPY, dtype=torch.float64, device="cuda:0").normal_()
dst = torch.empty_like(src)
for _ in range(5):
    eng.lib.fiveeq_stream_copy_f64(COPY, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), eng._stream())
torch.cuda.synchronize()
for _ in range(3):
    eng.reset_state()
    eng.run(mode="fused")
    torch.cuda.synchronize()
print("fused traffic workload done; algorithmic bytes per launch", eng.bytes_per_member_step("fused") * N * 750)
