#!/usr/bin/env python3
"""    python3 tools/fuzz_forms.py [seed] [seconds]
Randomised cross-check of every launch form / schedule / cache policy against the plain single-stream per-step run (bit for bit),
and of that run against the C oracle (1e-10) on a sample — for a few minutes."""
import os, sys, time, ctypes
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fiveeqscm_amd import _capi, emissions, params
from fiveeqscm_amd.engine import EnsembleEngine
from oracle import c_oracle
lib = _capi.load()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
t_end = time.time() + budget
LAYOUTS = {"co2": 1, "multigas": 3}
n_cases = n_runs = 0
while time.time() < t_end:
    kind = rng.choice(["co2", "multigas"])
    G = LAYOUTS[kind]
    dt = torch.float64 if rng.random() < 0.6 else torch.float32
    N = int(rng.choice([1, 2, 63, 64, 65, 255, 257, 1000, 4097, 16385, 70001, 70002, 300_001, 1_200_003]))
    n_steps = int(rng.integers(3, 90))
    t0 = int(rng.integers(0, 600))
    E = emissions.rcp_like_emissions(750, G)[t0:t0 + n_steps]
    p = params.sample_ensemble(params.default_params(kind), N, seed=int(rng.integers(1, 1 << 30)))
    out_steps = None if rng.random() < 0.5 else sorted(set(int(x) for x in rng.integers(0, n_steps, size=int(rng.integers(1, 6)))))
    stats = bool(rng.random() < 0.5)
    hist = None
    if rng.random() < 0.3:                                  # in-loop histograms of every step: modes 'fused', 'per_step', 'auto' only
        lo_h = float(rng.uniform(-2.0, 0.5))
        hist = (lo_h, lo_h + float(rng.uniform(0.5, 12.0)), int(rng.choice([1, 7, 512, 4096])))
    kw = dict(dtype=dt, device="cuda:0", output_steps=out_steps, collect_stats=stats, hist=hist)
    ref = EnsembleEngine(p, N, E, per_step_streams=1, chunk_members=0, **kw)
    lib.fiveeq_set_row_policy(0)
    ref.run(mode="per_step")
    torch.cuda.synchronize()
    names = ["R", "S"] + (["C", "T"] if ref.T is not None else []) + (["T_stats"] if stats else []) + (["T_hist"] if hist else [])
    if hist and out_steps is None:                          # ... equal the histograms of the stored rows
        assert torch.equal(ref.T_hist, ref.T_histogram(*hist)), ("T_hist vs stored rows", kind, N, str(dt), hist)
    want = {k: getattr(ref, k).clone() for k in names}
    if dt == torch.float64 and N <= 70002:
        o = c_oracle.run(E, p, N, n_threads=8)
        rows = ref.out_steps
        if ref.T is not None:
            got, exp = ref.T.cpu().numpy(), o["T"][rows]
            assert np.all(np.abs(got - exp) <= 1e-10 * np.abs(exp) + 1e-13), ("oracle T", kind, N)
            got, exp = ref.C.cpu().numpy(), o["C"][rows]
            assert np.all(np.abs(got - exp) <= 1e-10 * np.abs(exp) + 1e-13), ("oracle C", kind, N)
    ref.close()
    for trial in range(4):
        policy = int(rng.choice([0, 1, 2]))
        lib.fiveeq_set_row_policy(policy)
        streams = int(rng.choice([1, 2, 3]))
        chunk = int(rng.choice([0, 256, 1024, 65536])) if N > 300 else 0
        eng = EnsembleEngine(p, N, E, per_step_streams=streams, chunk_members=chunk, **kw)
        t, plan = 0, []
        while t < n_steps:
            seg = int(rng.integers(1, n_steps - t + 1))
            modes = ["per_step", "fused", "auto"] if hist else ["per_step", "graph", "fused", "ksteps", "auto"] + (["small"] if eng.small_form() else [])
            mode = str(rng.choice(modes))
            if mode == "per_step" and rng.random() < 0.3 and seg <= 4 and not hist:
                for tt in range(t, t + seg):
                    eng.step(tt)
                plan.append(("step", seg))
            else:
                eng.run(t, t + seg, mode=mode, join=bool(rng.random() < 0.7) or mode != "per_step")
                plan.append((mode, seg))
            t += seg
        if eng._ps_unjoined:
            eng.join()
        torch.cuda.synchronize()
        for k in names:
            got = getattr(eng, k)
            if k == "T_stats":      # sums: each form folds a wave's 64 values in its own order (rounding); min / max: exact
                ok = torch.equal(got[..., 2:], want[k][..., 2:]) and torch.allclose(got[..., :2], want[k][..., :2], rtol=1e-12, atol=1e-300)
            else:
                ok = torch.equal(got, want[k])
            assert ok, (k, kind, N, str(dt), n_steps, out_steps, stats, hist, policy, streams, chunk, plan)
        eng.close(); del eng
        n_runs += 1
    n_cases += 1
    del want
lib.fiveeq_set_row_policy(2)
print(f"fuzz ok: {n_cases} cases, {n_runs} randomised runs, all equal to the plain per-step run bit for bit")
