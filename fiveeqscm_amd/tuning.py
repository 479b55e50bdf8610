"""Box-dependent figures of the engine's schedules: how they are read from the environment and how they are measured.

The figures themselves (HBM_STREAM_BYTES_PER_S, LAUNCH_BOUNDARY_S, ...) are attributes of fiveeqscm_amd.engine, where the
schedules that use them live; `calibrate()` measures the two that depend on the box and writes them there."""
import os
import threading


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
# candidate take the time of one when the streams are concurrent and of two when they are not.
#
# WHEN (round 6): the probe synchronises both streams and clocks them on the wall, so it runs where the caller is synchronous
# anyway — EnsembleEngine.__init__ (for the stream current at construction) and EnsembleEngine.probe_streams(stream) — never
# inside run() / join() / a graph launch: those take what the cache holds for (device, caller's stream) and, for a stream nobody
# probed, plain unprobed side streams (correctness never depends on the overlap).  The verdict is RECORDED with the streams
# (`side_stream_report`): a probe that no candidate passed — several processes sharing the card, a profiler that serialises
# kernels — is visible on the engine and in bench.py's line instead of silently giving the serialised form.  FIVEEQ_SIDE_STREAM_PROBE=0
# skips probing altogether (counter-collection runs: no fiveeq_busy dispatches in the trace); a capturing stream is never probed.
_SIDE_STREAMS = {}               # (device index, hipStream handle) -> {"main", "streams", "probed", "passed", "tried"}
_SIDE_LOCK = threading.RLock()   # the C ABI is used from several host threads: the cache is theirs in common
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


def probe_enabled():
    """FIVEEQ_SIDE_STREAM_PROBE=0 switches the probe off (plain side streams, recorded as unprobed)."""
    return _env_choice("FIVEEQ_SIDE_STREAM_PROBE", "1", ("0", "1")) == "1"


def _capturing(stream):
    import torch
    with torch.cuda.stream(stream):
        return bool(torch.cuda.is_current_stream_capturing())


def concurrent_side_streams(lib, main, count, probe=False):
    """`count` HIP streams on main's device to run beside `main`, shared by every caller in the process (one entry per
    (device, main)).

    probe=False (run(), join(), graph launches): never synchronises — the cached streams, topped up with plain unprobed ones.
    probe=True (engine construction, EnsembleEngine.probe_streams): entries that were never probed are (re)built from candidates
    that pass `streams_concurrent` against main AND against each other; a candidate that fails is dropped (torch recycles its
    pool of streams, so nothing leaks); if none of _PROBE_CANDIDATES passes, the last one is taken as it is and the entry records
    passed=False.  Not probed at all: FIVEEQ_SIDE_STREAM_PROBE=0, a capturing `main`, an ExternalStream (its handle may be
    destroyed and reused by the owner, so a verdict about it is not worth caching)."""
    import torch
    key = (main.device.index, main.cuda_stream)
    with _SIDE_LOCK:
        ent = _SIDE_STREAMS.get(key)
        if ent is None:
            # `main` is kept: a torch-pooled stream object alive here keeps its handle from being handed to anyone else
            ent = _SIDE_STREAMS[key] = {"main": main, "streams": [], "probed": False, "passed": [], "tried": 0}
        may_probe = (probe and probe_enabled() and not isinstance(main, torch.cuda.ExternalStream) and not _capturing(main))
        if may_probe and not (ent["probed"] and len(ent["streams"]) >= count):
            with torch.cuda.device(main.device):
                scratch = torch.zeros(2, dtype=torch.float64, device=main.device)
                have = ent["streams"] if ent["probed"] else []
                passed = ent["passed"] if ent["probed"] else []
                while len(have) < count:
                    cand, ok = None, False
                    for _ in range(_PROBE_CANDIDATES):
                        cand = torch.cuda.Stream(device=main.device)
                        ent["tried"] += 1
                        ok = all(streams_concurrent(lib, s, cand, scratch) for s in [main] + have)
                        if ok:
                            break
                    have.append(cand)
                    passed.append(bool(ok))
                ent.update(streams=have, passed=passed, probed=True)
        while len(ent["streams"]) < count:                         # unprobed top-up: no synchronisation, no verdict
            with torch.cuda.device(main.device):
                ent["streams"].append(torch.cuda.Stream(device=main.device))
            ent["passed"].append(None)
        return list(ent["streams"][:count])


def side_stream_report(main, count=None):
    """What the cache knows about main's side streams: {"probed", "passed" (True / False per stream, None = not probed),
    "candidates_tried", "probe_enabled"} — on the engine as `side_stream_report()`, in bench.py's line as config.side_streams."""
    with _SIDE_LOCK:
        ent = _SIDE_STREAMS.get((main.device.index, main.cuda_stream))
        if ent is None:
            return {"probed": False, "passed": [], "candidates_tried": 0, "probe_enabled": probe_enabled()}
        passed = list(ent["passed"] if count is None else ent["passed"][:count])
        return {"probed": bool(ent["probed"]), "passed": passed, "candidates_tried": int(ent["tried"]),
                "probe_enabled": probe_enabled()}
