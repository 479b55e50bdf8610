#!/usr/bin/env python3
"""Experiment: do two member chunks on two HIP streams (kernels of different chunks may overlap,
filling each other's launch ramp / tail) beat one stream?  Per-step path, fp64 multigas."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fiveeqscm_amd import emissions, params
from fiveeqscm_amd.engine import EnsembleEngine

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
steps = 300
base = params.sample_ensemble(params.default_params("multigas"), 65536)
p = dict(base)
for k in ("r0", "rC", "rT", "q"):
    p[k] = np.tile(base[k], (1, -(-N // 65536)))[:, :N]
eng = EnsembleEngine(p, N, emissions.rcp_like_emissions(steps + 20, 3), device="cuda:0", chunk_members=None)
lib = eng.lib
fn = lib.fiveeq_run_f64
streams = [torch.cuda.Stream() for _ in range(4)]


def run_split(t0, t1, parts):
    c = -(-N // parts // 256) * 256
    for i, m0 in enumerate(range(0, N, c)):
        n = min(c, N - m0)
        rc = fn(*eng._run_args(t0, t1, m0, n), ctypes.c_void_p(streams[i % len(streams)].cuda_stream))
        assert rc == 0


def timeit(f):
    f(0, 20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f(20, 20 + steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / steps


ref = timeit(lambda a, b: eng.run(a, b))
Tref = eng.T.clone()
print(f"N={N}: one stream {ref:.2f} us/step")
for parts in (2, 3, 4):
    eng.reset_state()
    t = timeit(lambda a, b: run_split(a, b, parts))
    print(f"  {parts} chunks on {parts} streams: {t:.2f} us/step   identical={torch.equal(eng.T[:320], Tref[:320])}")
