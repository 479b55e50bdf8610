#!/usr/bin/env python3
"""Experiment (needs a build with -DFIVEEQ_FUSED_TIMING): when and on which CU does every wave of ONE launch of the fused
packed-fp32 kernel run?  Prints the number of waves resident per CU over time and wave lifetimes by start order.
    python3 tools/fused_timing.py /tmp/fiveeq_variants/libfiveeq_FT.so [workgroups_per_cu ...]"""
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
STEPS = 200
E = emissions.rcp_like_emissions(750, 3)[:STEPS]
for k in [int(v) for v in sys.argv[2:]] or [4, 8]:
    N = 256 * k * 512
    p = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0", dtype=torch.float32)
    eng = EnsembleEngine(p, N, E, dtype=torch.float32, store_trajectory=False, collect_stats=True, lib_path=lib)
    for rep in range(2):
        eng.reset_state()
        eng.run(mode="fused")
        torch.cuda.synchronize()
    rec = eng.T_stats[0::2, STEPS - 1, :].cpu().numpy()           # one record per wave (packed lanes: records 2w, 2w+1)
    t0, t1 = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64)
    hw, xcc = rec[:, 2].astype(np.int64), rec[:, 3].astype(np.int64)
    base = t0.min()
    start, end = (t0 - base) * 0.01, (t1 - base) * 0.01            # us
    life = end - start
    cu = ((xcc & 0xf) << 12) | (((hw >> 13) & 0x7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
    simd = (hw >> 4) & 0x3
    n_cu = len(set(cu.tolist()))
    print(f"== {k} workgroups of 256 per CU on average ({N} members, {len(rec)} waves, {STEPS} steps): kernel span {end.max():.1f} us, "
          f"{end.max() / STEPS:.3f} us/step; {n_cu} distinct CUs")
    per_cu = collections.Counter(cu.tolist())
    print("   waves per CU over the whole launch: min/median/max", min(per_cu.values()), int(np.median(list(per_cu.values()))), max(per_cu.values()))
    # resident waves per CU at sample times
    for frac in (0.05, 0.25, 0.5, 0.75, 0.95):
        t = end.max() * frac
        live = (start <= t) & (end > t)
        c = collections.Counter(cu[live].tolist())
        cs = collections.Counter((a, b) for a, b in zip(cu[live].tolist(), simd[live].tolist()))
        vals = list(c.values()) + [0] * (n_cu - len(c))
        print(f"   t = {t:8.1f} us: resident waves {int(live.sum()):5d}; per CU min/median/max {min(vals)}/{int(np.median(vals))}/{max(vals)}; "
              f"per SIMD max {max(cs.values()) if cs else 0}")
    order = np.argsort(start)
    q = len(order) // 4
    print(f"   wave lifetime (us) by start order, quartiles: " + "  ".join(f"{np.median(life[order[i * q:(i + 1) * q]]):.1f}" for i in range(4)),
          f"  start of the last wave {start.max():.1f}")
    first = start < 5.0
    print(f"   waves that start in the first 5 us: {int(first.sum())}; their lifetime median {np.median(life[first]):.1f}; the others' "
          f"{np.median(life[~first]) if (~first).any() else float('nan'):.1f}")
    del eng, p
