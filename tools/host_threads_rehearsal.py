#!/usr/bin/env python3
"""EIGHT host threads enqueuing at once — in ONE process, because the GPU pool allows six processes on a card and the launcher's
agent is one of them (tools/rehearse_multi_gpu.sh stops at five ranks).  Each thread owns an engine of the driver's workload
(1M fp64 members; no stored trajectory, so eight of them fit the card) and its own HIP stream pair, waits at a common barrier
and enqueues a burst of 20 per-step timesteps on a drained device, 15 times; reported: the median enqueue time per step of the
SLOWEST thread.  Threads of one process share the GIL (the ctypes launch calls release it) and the HIP runtime's locks, which
8 processes on an 8-GPU node do not: this is an UPPER bound on what contention can cost the host side there.
    python3 tools/host_threads_rehearsal.py [threads ...]
Round 6: under a launcher (python -m torch.distributed.run --nproc-per-node P ... tools/host_threads_rehearsal.py T) every one of
the P processes runs T such threads and all P x T bursts start behind a common gloo barrier: 4 processes x 2 threads are EIGHT
enqueuing threads in four GPU processes — as close to eight ranks as the pool's six-process limit allows, with the GIL shared
by pairs only.  Each process binds itself to its slice of the GPU's CPUs first (fiveeqscm_amd/hostbind.py), like a bench rank."""
import os
import sys
import threading
import time

import numpy as np

ROOT_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT_)
WORLD, RANK = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
if WORLD > 1:                                          # before anything touches the GPU
    from fiveeqscm_amd import hostbind
    BIND = hostbind.bind_rank(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", str(WORLD))))
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

N, K, REPS = 1_000_000, 20, 15
dev = torch.device("cuda:0")
dist = None
if WORLD > 1:
    import torch.distributed as dist
    dist.init_process_group("gloo")
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
if RANK == 0:
    print(f"{WORLD} process(es); " + (f"rank 0 bound to CPUs {BIND.get('cpus')} (applied {BIND.get('applied')})" if WORLD > 1 else "unbound"))
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
                if dist is not None:
                    dist.barrier()                       # every process's threads start their burst together
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
    slow, fast = max(out), min(out)
    if dist is not None:
        tt = torch.tensor([slow, -fast], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        slow, fast = float(tt[0]), -float(tt[1])
    if RANK == 0:
        print(f"  {WORLD} x {n} = {WORLD * n} thread(s) enqueuing at once: slowest thread {slow:6.2f} us per timestep (fastest {fast:6.2f}) "
              f"= {slow / 34.5:.2f} of an un-shared 34.5 us step")
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
