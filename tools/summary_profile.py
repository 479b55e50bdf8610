#!/usr/bin/env python3
"""cProfile of one warm gather_summary call on 3 x 12.5M fp32 device rows: where the host time of the summary goes."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd.distributed import gather_summary  # noqa: E402

x = torch.randn((3, 12_500_000), device="cuda:0", dtype=torch.float32) * 0.7 + 2.0
for _ in range(3):
    gather_summary(x)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    gather_summary(x)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(35)
