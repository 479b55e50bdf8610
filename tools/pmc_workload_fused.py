#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc passes on the time-fused kernel: one 300-step launch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dtype = torch.float32 if (len(sys.argv) > 2 and sys.argv[2] == "f32") else torch.float64
base = params.sample_ensemble(params.default_params("multigas"), 65536)
p = dict(base)
for k in ("r0", "rC", "rT", "q"):
    p[k] = np.tile(base[k], (1, -(-N // 65536)))[:, :N]
eng = EnsembleEngine(p, N, emissions.rcp_like_emissions(330, 3), device="cuda:0", dtype=dtype)
eng.run(0, 20, mode="fused")
eng.run(20, 320, mode="fused")
torch.cuda.synchronize()
print("done")
