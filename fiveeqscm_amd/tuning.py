"""Box-dependent figures of the engine's schedules: how they are read from the environment and how they are measured.

The figures themselves (HBM_STREAM_BYTES_PER_S, LAUNCH_BOUNDARY_S, ...) are attributes of fiveeqscm_amd.engine, where the
schedules that use them live; `calibrate()` measures the two that depend on the box and writes them there."""
import os


def _env_positive(name, default):
    """A positive float from the environment; anything else (empty, garbage, zero, negative, nan) keeps the default, with a warning."""
    raw = os.environ.get(name)
    if raw is None:
        return default
    try:
        val = float(raw)
    except ValueError:
        val = float("nan")
    if not (val > 0.0) or val == float("inf"):
        import warnings
        warnings.warn(f"{name}={raw!r} is not a positive number: using the default {default:g}")
        return default
    return val


def _env_choice(name, default, choices):
    raw = os.environ.get(name, default)
    if raw not in choices:
        raise ValueError(f"{name}={raw!r}: must be one of {sorted(choices)}")
    return raw


def calibrate(device="cuda:0", members=1_000_000, apply=True):
    """Measure the two box-dependent figures on `device` with the per-step kernel itself and (apply=True) make them the
    module's HBM_STREAM_BYTES_PER_S / LAUNCH_BOUNDARY_S: the dependent-launch boundary as the time per step of a 64-member
    ensemble (nothing but launches), the streaming ceiling as algorithmic bytes per second of a per-step run of `members`
    fp64 members, one launch per step on one stream.  Returns {"launch_boundary_s", "hbm_stream_bytes_per_s"}."""
    import time

    import torch

    from . import emissions, engine, params
    EnsembleEngine = engine.EnsembleEngine
    E = emissions.rcp_like_emissions(200, 3)
    out = {}
    for key, n in (("launch_boundary_s", 64), ("hbm_stream_bytes_per_s", int(members))):
        p = params.sample_ensemble_shard(params.default_params("multigas"), n, device=device)
        eng = EnsembleEngine(p, n, E, device=device, store_trajectory=False, chunk_members=None, per_step_streams=1)
        best = None
        for _ in range(3):
            eng.reset_state()
            torch.cuda.synchronize(eng.device)
            t0 = time.perf_counter()
            eng.run(mode="per_step")
            torch.cuda.synchronize(eng.device)
            dt = (time.perf_counter() - t0) / eng.n_steps
            best = dt if best is None else min(best, dt)
        out[key] = best if n == 64 else n * eng.bytes_per_member_step("per_step") / best
        eng.close()
        del eng, p
    if apply:
        engine.LAUNCH_BOUNDARY_S, engine.HBM_STREAM_BYTES_PER_S = out["launch_boundary_s"], out["hbm_stream_bytes_per_s"]
    return out
