#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: 20 warm gather_summary calls on 3 x 12.5M device rows (fp32, then fp64),
so that the per-kernel average durations of the four summary passes can be read off the stats file."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd.distributed import gather_summary  # noqa: E402

for dt in (torch.float32, torch.float64):
    x = torch.randn((3, 12_500_000), device="cuda:0", dtype=dt) * 0.7 + 2.0
    for _ in range(20):
        gather_summary(x)
    torch.cuda.synchronize()
