#!/usr/bin/env python3
"""EIGHT host threads enqueuing at once — in ONE process, because the GPU pool allows six processes on a card and the launcher's
agent is one of them (tools/host_share_rehearsal.sh stops at five ranks).  Each thread owns an engine of the driver's workload
(1M fp64 members; no stored trajectory, so eight of them fit the card) and its own HIP stream pair, waits at a common barrier
and enqueues a burst of 20 per-step timesteps on a drained device, 15 times; reported: the median enqueue time per step of the
SLOWEST thread.  Threads of one process share the GIL (the ctypes launch calls release it) and the HIP runtime's locks, which
8 processes on an 8-GPU node do not: this is an UPPER bound on what contention can cost the host side there.
    python3 tools/host_threads_rehearsal.py [threads ...]"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N, K, REPS = 1_000_000, 20, 15
dev = torch.device("cuda:0")
E = emissions.rcp_like_emissions(750, 3)
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device=dev)
counts = [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]
engines = []
for i in range(max(counts)):
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        eng = EnsembleEngine(p, N, E, device=dev, store_trajectory=False)
        eng.run(0, 30, stream=s)
    engines.append((eng, s))
torch.cuda.synchronize()
print(f"{N} fp64 members per thread, bursts of {K} per-step timesteps ({engines[0][0].per_step_streams} launches per timestep), "
      f"median of {REPS} bursts, every burst behind a common barrier on a drained device")
for n in counts:
    bar = threading.Barrier(n)
    out = [None] * n

    def work(i):
        eng, s = engines[i]
        mine = []
        for r in range(REPS):
            if i == 0:
                torch.cuda.synchronize()
            bar.wait()
            t = (r * K) % 700
            t0 = time.perf_counter()
            eng.run(t, t + K, stream=s, join=False)
            mine.append((time.perf_counter() - t0) / K * 1e6)
            eng.join(s)
            bar.wait()
        out[i] = float(np.median(mine))

    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    torch.cuda.synchronize()
    print(f"  {n} thread(s) enqueuing at once: slowest thread {max(out):6.2f} us per timestep (fastest {min(out):6.2f}) "
          f"= {max(out) / 34.5:.2f} of an un-shared 34.5 us step")
