#!/usr/bin/env python3
"""Experiment (needs a build with -DFIVEEQ_TILE_TIMING): start / end time and placement of every persistent workgroup of ONE
launch of the time-tiled kernel.    python3 tools/tile_timing.py /tmp/fiveeq_variants/libfiveeq_TT.so [members] [K]"""
import collections
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402

lib = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12_500_000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 11
p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=torch.float32)
E = emissions.rcp_like_emissions(750, 3)
eng = EnsembleEngine(p, N, E, dtype=torch.float32, store_trajectory=False, collect_stats=True, hist=(-2.0, 12.0, 4096), lib_path=lib)
for rep in range(3):
    eng.reset_state()
    eng.run(t_begin=0, t_end=K, mode="tiled", k_steps=K)
    torch.cuda.synchronize()
    row = eng.T_hist[K - 1].cpu().numpy().astype(np.uint64)
    n_wg = 256
    rec = row[:n_wg * 4].reshape(n_wg, 4)
    t0, t1, hw, xcc = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64), rec[:, 2], rec[:, 3]
    base = t0.min()
    dur = (t1 - t0) * 0.01          # us (100 MHz)
    start = (t0 - base) * 0.01
    end = (t1 - base) * 0.01
    print(f"rep {rep}: kernel span {end.max():.1f} us; workgroup duration min/median/max {dur.min():.1f}/{np.median(dur):.1f}/{dur.max():.1f} us; "
          f"start spread {start.max():.1f} us; end min/median/max {end.min():.1f}/{np.median(end):.1f}/{end.max():.1f}")
    if rep == 2:
        cu = (hw >> np.uint64(8)) & np.uint64(0xf)
        sh = (hw >> np.uint64(12)) & np.uint64(0x1)
        se = (hw >> np.uint64(13)) & np.uint64(0x7)
        x = xcc & np.uint64(0xf)
        place = collections.Counter(zip(x.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
        print("distinct (xcc, se, sh, cu) places:", len(place), "  workgroups sharing a place:", sum(1 for v in place.values() if v > 1))
        byx = collections.defaultdict(list)
        for i in range(n_wg):
            byx[int(x[i])].append(dur[i])
        for k in sorted(byx):
            print(f"  xcc {k}: {len(byx[k])} workgroups, duration median {np.median(byx[k]):.1f} max {max(byx[k]):.1f}")
        order = np.argsort(dur)
        print("  slowest 8:", [(int(i), round(float(dur[i]), 1), int(x[i]), int(se[i]), int(cu[i])) for i in order[-8:]])
        print("  fastest 8:", [(int(i), round(float(dur[i]), 1), int(x[i]), int(se[i]), int(cu[i])) for i in order[:8]])
        hist, edges = np.histogram(dur, bins=12)
        print("  duration histogram:", list(zip(np.round(edges[:-1], 1).tolist(), hist.tolist())))
