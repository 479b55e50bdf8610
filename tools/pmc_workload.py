#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes: a few calibration copies (known bytes) followed by
per-step launches of the bench workload.
Run as `rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- python3 tools/pmc_workload.py [members] [kind]`."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "multigas"
COPY = 1 << 27
G = 3 if kind == "multigas" else 1
base = params.sample_ensemble(params.default_params(kind), 65536)
p = dict(base)
for k in ("r0", "rC", "rT", "q"):
    p[k] = np.tile(base[k], (1, -(-N // 65536)))[:, :N]
eng = EnsembleEngine(p, N, emissions.rcp_like_emissions(750, G), device="cuda:0")
src = torch.empty(COPY, dtype=torch.float64, device="cuda:0").normal_()
dst = torch.empty_like(src)
for _ in range(5):
    eng.lib.fiveeq_stream_copy_f64(COPY, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), eng._stream())
torch.cuda.synchronize()
eng.run(0, 60)
torch.cuda.synchronize()
print("pmc workload done", N, kind, COPY)
