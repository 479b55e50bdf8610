#!/usr/bin/env python3
"""Where does the host time of a per-step burst go?  cProfile of eng.run(t, t + K, join=False) on a drained device (K = 20, the
driver's block), and the bare C call beside it.     python3 tools/host_enqueue_profile.py [members] [K]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device=dev)
eng = EnsembleEngine(p, N, emissions.rcp_like_emissions(750, 3), device=dev)
eng.run(0, 50)
torch.cuda.synchronize()


def burst(reps, k, prof=None):
    tot = 0.0
    for i in range(reps):
        t = (i * k) % (750 - k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if prof is not None:
            prof.enable()
        eng.run(t, t + k, join=False)
        if prof is not None:
            prof.disable()
        tot += time.perf_counter() - t0
        eng.join()
    torch.cuda.synchronize()
    return tot / (reps * k) * 1e6


print(f"{N} members, bursts of {K} steps: {burst(50, K):.2f} us/step host enqueue; bursts of 200: {burst(10, 200):.2f} us/step")
pr = cProfile.Profile()
burst(100, K, pr)
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(18)
