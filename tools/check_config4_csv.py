#!/usr/bin/env python3
"""Check the CSV example/run_sharded.py wrote for BASELINE configs[3] (10M members over R ranks) against ONE engine running the
whole ensemble and NumPy on its rows: the exchanged percentiles must equal np.percentile of the single run BIT FOR BIT (the Latin
hypercube and the summary do not depend on the world size; members never interact), the moments to rounding.
    python3 tools/check_config4_csv.py summary.csv 10000000 249,499,749"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import emissions, params, scenario  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402


def main():
    path, N, steps = sys.argv[1], int(sys.argv[2]), [int(s) for s in sys.argv[3].split(",")]
    years, cols = scenario.read_summary_csv(path)
    whole = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0")
    eng = EnsembleEngine(whole, N, emissions.rcp_like_emissions(750, 3), device="cuda:0", output_steps=steps, store_concentrations=False)
    eng.run(mode="fused")                                   # another launch form than the ranks' (bit-identical by construction)
    torch.cuda.synchronize()
    T = eng.T.cpu().numpy()
    want = np.percentile(T, (5.0, 50.0, 95.0), axis=1).T
    ok_pct = np.array_equal(cols["percentiles"], want)
    ok_cnt = cols["count"].tolist() == [float(N)] * len(steps)
    ok_mean = np.allclose(cols["mean"], T.mean(1), rtol=1e-13, atol=0)
    ok_mm = np.array_equal(cols["min"], T.min(1)) and np.array_equal(cols["max"], T.max(1))
    print(f"{path}: years {years.tolist()}, {N} members")
    for k, t in enumerate(steps):
        print(f"  step {t}: p05/p50/p95 exchanged {cols['percentiles'][k].tolist()}  np.percentile of one engine's rows {want[k].tolist()}")
    print(f"percentiles bit for bit: {ok_pct}; count: {ok_cnt}; mean to 1e-13: {ok_mean}; min / max exact: {ok_mm}")
    sys.exit(0 if ok_pct and ok_cnt and ok_mean and ok_mm else 1)


if __name__ == "__main__":
    main()
