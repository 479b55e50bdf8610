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


# ---- side streams that really run beside the caller's stream ------------------------------------------------------------------
# HIP maps its streams onto a handful of hardware queues (four by default) in creation order, and two streams on the same queue
# run one after the other whatever the program says.  A schedule that overlaps two launches (the two member parts of a per-step
# launch, the histogram pass beside the next fused chunk) then silently loses its overlap: the per-step form is 12 % slower
# (4M fp64 members: 158 us per step instead of 141) for every fourth stream torch hands out (profiles/r05/side_stream_probe.txt).
# So the side streams are PROBED: two launches of known duration (fiveeq_busy, one wave each) on the caller's stream and on the
# candidate take the time of one when the streams are concurrent and of two when they are not.  The answer depends on the pair of
# streams only: it is cached per (device, caller's stream) and shared by every engine of the process.
_SIDE_STREAMS = {}
_PROBE_ITERS = 60_000            # ~0.2 ms per launch: far above the launch and synchronisation overheads around it
_PROBE_CANDIDATES = 8


def _pair_time(lib, a, b, scratch, iters):
    """Wall time (s) of one busy launch on stream a and — b given — another on b, both behind the same start."""
    import ctypes
    import time
    a.synchronize()
    if b is not None:
        b.synchronize()
    ptr = lambda i: ctypes.c_void_p(scratch.data_ptr() + 8 * i)   # noqa: E731
    t0 = time.perf_counter()
    lib.fiveeq_busy(iters, ptr(0), ctypes.c_void_p(a.cuda_stream))
    if b is not None:
        lib.fiveeq_busy(iters, ptr(1), ctypes.c_void_p(b.cuda_stream))
        b.synchronize()
    a.synchronize()
    return time.perf_counter() - t0


def streams_concurrent(lib, a, b, scratch=None):
    """True when launches on HIP streams a and b (torch.cuda.Stream) overlap on this box right now: best of three timings of a
    pair of one-wave launches against best of three of a single one.  Synchronises both streams."""
    import torch
    if a.cuda_stream == b.cuda_stream:
        return False
    with torch.cuda.device(a.device):
        if scratch is None:
            scratch = torch.zeros(2, dtype=torch.float64, device=a.device)
        _pair_time(lib, a, b, scratch, 1000)                                   # code object load, clocks
        one = min(_pair_time(lib, a, None, scratch, _PROBE_ITERS) for _ in range(3))
        two = min(_pair_time(lib, a, b, scratch, _PROBE_ITERS) for _ in range(3))
    return two < 1.5 * one


def concurrent_side_streams(lib, main, count):
    """`count` HIP streams on main's device that run beside `main` AND beside each other, probed once per (device, main) and
    shared by every caller in the process.  A candidate that fails the probe is dropped (torch recycles its pool of streams, so
    nothing leaks); if no candidate passes within _PROBE_CANDIDATES tries the last one is taken as it is — correctness never
    depends on the overlap."""
    import torch
    key = (main.device.index, main.cuda_stream)
    have = _SIDE_STREAMS.setdefault(key, [])
    if len(have) < count:
        with torch.cuda.device(main.device):
            scratch = torch.zeros(2, dtype=torch.float64, device=main.device)
            while len(have) < count:
                cand = None
                for _ in range(_PROBE_CANDIDATES):
                    cand = torch.cuda.Stream(device=main.device)
                    if all(streams_concurrent(lib, s, cand, scratch) for s in [main] + have):
                        break
                have.append(cand)
    return have[:count]
