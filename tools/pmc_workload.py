#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: a few calibration copies (known bytes) followed by
per-step launches of the bench workload.
Run as `rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 tools/pmc_workload.py [members] [kind] [f64|f32]`."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "multigas"
dt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.float64
COPY = 1 << 27
G = 3 if kind == "multigas" else 1
p = params.sample_ensemble_shard(params.default_params(kind), N, device="cuda:0", dtype=dt)      # drawn on the device
eng = EnsembleEngine(p, N, emissions.rcp_like_emissions(750, G), device="cuda:0", dtype=dt)
src = torch.empty(COPY, dtype=torch.float64, device="cuda:0").normal_()
dst = torch.empty_like(src)
for _ in range(5):
    eng.lib.fiveeq_stream_copy_f64(COPY, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), eng._stream())
torch.cuda.synchronize()
eng.run(0, 60)
torch.cuda.synchronize()
print("pmc workload done", N, kind, COPY)
