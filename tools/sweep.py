#!/usr/bin/env python3
"""Exploration tool (not part of the product): time engine modes over ensemble sizes, and A/B
several builds of libfiveeq_hip.so in ONE process with interleaved rounds
(cdna_hip_programming.md section 5.4 rule 24).

    python tools/sweep.py --members 250000,1000000,4000000 --modes per_step,fused
    python tools/sweep.py --libs fiveeqscm_amd/csrc/libfiveeq_hip.so,/path/variant.so --rounds 5
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", default="1000000")
    ap.add_argument("--modes", default="per_step,fused")
    ap.add_argument("--kind", default="multigas")
    ap.add_argument("--dtypes", default="f64")
    ap.add_argument("--libs", default="")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--no-trajectory", action="store_true")
    ap.add_argument("--stats", action="store_true", help="collect on-device T statistics")
    ap.add_argument("--scenario-steps", type=int, default=750, help="trajectory capacity (memory = steps*(G+1)*N*w)")
    a = ap.parse_args()
    G = 3 if a.kind == "multigas" else 1
    libs = [x for x in a.libs.split(",") if x] or [None]
    E = emissions.rcp_like_emissions(a.scenario_steps, G)
    print(f"{'lib':28s} {'dtype':5s} {'members':>9s} {'mode':9s} {'us/step(med)':>12s} {'us/step(min)':>12s} "
          f"{'Gmember-steps/s':>15s} {'alg GB/s':>9s}")
    for N in [int(x) for x in a.members.split(",")]:
        p = params.sample_ensemble_shard(params.default_params(a.kind), N, device="cuda:0")   # drawn on the device
        for dt in a.dtypes.split(","):
            dtype = torch.float64 if dt == "f64" else torch.float32
            # ONE set of device buffers; the builds under test are swapped in as `eng.lib`.  (Separate
            # engines per build showed up to 4.5 % spread between IDENTICAL libraries, purely from where
            # their 24 GB of buffers landed.)
            pd = dict(p)
            for k in ("r0", "rC", "rT", "q"):
                pd[k] = p[k].to(dtype)
            eng = EnsembleEngine(pd, N, E, dtype=dtype, device="cuda:0", lib_path=libs[0],
                                 store_trajectory=not a.no_trajectory, collect_stats=a.stats)
            from fiveeqscm_amd import _capi
            handles = {lib: _capi.load(lib) for lib in libs}
            for mode in a.modes.split(","):
                times = {lib: [] for lib in libs}
                for rnd in range(a.rounds + 1):
                    for lib in (libs if rnd % 2 == 0 else libs[::-1]):
                        eng.lib = handles[lib]
                        eng.close()                               # plans belong to the previous library
                        eng.reset_state()
                        eng.run(0, 20, mode=mode)
                        torch.cuda.synchronize()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        eng.run(20, 20 + a.steps, mode=mode)
                        e1.record()
                        e1.synchronize()
                        if rnd:
                            times[lib].append(e0.elapsed_time(e1) * 1e3 / a.steps)
                for lib in libs:
                    t = np.array(times[lib])
                    med = float(np.median(t))
                    A = eng.bytes_per_member_step(mode)
                    print(f"{os.path.basename(lib or 'default'):28s} {dt:5s} {N:9d} {mode:9s} {med:12.2f} {t.min():12.2f} "
                          f"{N / med / 1e3:15.3f} {A * N / med / 1e3:9.0f}", flush=True)
            eng.close()
            del eng
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
