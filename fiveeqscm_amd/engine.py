"""Python host of the ensemble engine: torch-ROCm tensors own the memory, the C ABI
(include/fiveeq.h) does the work.  One engine = one device = one member shard.

Data layout in HBM (struct-of-arrays over members; N = members of this shard):
    r      [3G, N]   rows g*3 + (0,1,2) = r0, rC, rT of gas g
    q      [2,  N]   thermal-box coefficients
    R      [SP, N]   pool contents, gas-major (SP = sum of active pools)
    S      [2,  N]   thermal-box temperatures
    drive  [n_steps, 8]   shared: E_g, cumulative E_g before the step, F_ext, output row
    C      [n_rows, G, N]    concentrations of the stored steps (all steps, a selection, or none)
    T      [n_rows, N]       temperature of the stored steps
    T_stats[W, n_steps, 4]   optional per-wave (sum, sum^2, min, max) of T, W = ceil(N/64), fp64 (0.5 B per member-step:
                             allocated by the first run that writes wave records, not at construction)
    T_hist [n_steps, n_bins] optional fixed-bin histogram of T for EVERY step, accumulated inside the
                             time loop by the tiled kernel (run(mode="tiled")), int64

There is no CPU path: constructing an engine without a GPU, or without the built
HIP library, raises.  (The reference's own function, `calculate_hfc_conc`, is a
NumPy one-liner and lives in fiveeqscm_amd.concentrations.)
"""
import ctypes
import os

import numpy as np
import torch

from . import _capi
from .emissions import make_drive
from .params import make_model, n_gas_of, pools_of

_DTYPES = {torch.float64: "f64", torch.float32: "f32"}
INFINITY_CACHE_BYTES = 256 << 20     # MI355X die-level L3 (MI355X_MICROARCH.md); sizes the chunk-major schedule
# The two box-dependent figures the schedules are derived from.  The defaults are what round 1-3 measured on MI355X; another
# box (or a future driver) can set them from the environment, or measure them with `calibrate()` below, which overwrites these
# module attributes — engines created afterwards use the new values.



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


HBM_STREAM_BYTES_PER_S = _env_positive("FIVEEQ_HBM_STREAM_BYTES_PER_S", 6.7e12)    # ceiling of the per-step kernel
LAUNCH_BOUNDARY_S = _env_positive("FIVEEQ_LAUNCH_BOUNDARY_S", 2.0e-6)              # dependent-launch boundary on one stream (measured 1.5-2.6 us)
PER_STEP_SPLIT_MIN_S = 16.0e-6       # a per-step launch is split over two streams from this much traffic time on
PER_STEP_BLOCK = 25                  # steps enqueued per part before switching to the next part's stream
FUSED_SPAN_STEPS = 128               # mode='fused': steps per launch for ensembles of few rounds of waves (see fused_span)
FUSED_SPAN_MAX_ROUNDS = 8.0          # ... up to this many rounds of 4 waves per SIMD
FUSED_SPAN_MIN_ROUNDS = 0.25         # ... and from this many on


def _rows(x, K, N, name):
    if isinstance(x, torch.Tensor):                    # already on a device (params.sample_ensemble_shard)
        if tuple(x.shape) != (K, N):
            raise ValueError(f"{name}: tensor shape {tuple(x.shape)}, want [{K},{N}]")
        return x
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 0:
        x = x.reshape(1)
    if x.ndim == 1:
        if x.shape[0] != K:
            raise ValueError(f"{name}: shape {x.shape}, want [{K}] or [{K},{N}]")
        return np.broadcast_to(x[:, None], (K, N))
    if x.shape != (K, N):
        raise ValueError(f"{name}: shape {x.shape}, want [{K}] or [{K},{N}]")
    return x


def calibrate(device="cuda:0", members=1_000_000, apply=True):
    """Measure the two box-dependent figures on `device` with the per-step kernel itself and (apply=True) make them the
    module's HBM_STREAM_BYTES_PER_S / LAUNCH_BOUNDARY_S: the dependent-launch boundary as the time per step of a 64-member
    ensemble (nothing but launches), the streaming ceiling as algorithmic bytes per second of a per-step run of `members`
    fp64 members, one launch per step on one stream.  Returns {"launch_boundary_s", "hbm_stream_bytes_per_s"}."""
    import time

    from . import emissions, params
    global HBM_STREAM_BYTES_PER_S, LAUNCH_BOUNDARY_S
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
        LAUNCH_BOUNDARY_S, HBM_STREAM_BYTES_PER_S = out["launch_boundary_s"], out["hbm_stream_bytes_per_s"]
    return out


class EnsembleEngine:
    """Advance N ensemble members of the five-equation model on one MI355X."""

    def __init__(self, params, n_members, emissions, *, F_ext=None, dt=1.0, dtype=torch.float64,
                 device=None, store_trajectory=True, output_steps=None, store_concentrations=True,
                 collect_stats=False, hist=None, hist_ring_steps="auto", hist_ring="bins",
                 concentration_driven=False, chunk_members="auto", per_step_streams="auto", fused_span="auto", R0=None,
                 S0=None, lib_path=None):
        """store_trajectory / output_steps: True stores C, T of every step; a list of step indices
        stores only those (rows in increasing step order, see `out_steps`); False stores nothing.
        store_concentrations=False keeps only the T rows (a 100M-member fp32 run then stores 4 B instead
        of 16 B per member and stored step: all 750 steps fit, and `T_histogram` gives every step's percentiles).
        collect_stats: also accumulate per-step ensemble moments of T on the device (`stats()`).
        hist=(lo, hi, n_bins): allocate `T_hist` [n_steps, n_bins] (int64), the fixed-bin histogram of T of
        EVERY step, for all-timestep percentiles (distributed.histogram_percentiles) without a stored
        trajectory.  Three ways to fill it, same counts bit for bit:
          run(mode="fused")    streams it, `hist_ring_steps` steps at a time through a two-slot ring that a histogram kernel
                               drains on a second HIP stream while the next slot is computed.  hist_ring="bins" (default):
                               the fused kernel writes each member's BIN INDEX (uint16: 2 bytes per member-step; 3.2 GB for
                               12.5M members at S = 64) and keeps the statistics in the kernel; hist_ring="T": it parks T
                               itself (4 or 8 bytes) and the pass returns the per-step moments with the counts (no in-kernel
                               statistics; T rows only, no stored concentrations).  "auto" ring length: what 8 GB of T
                               ring would hold, at most 128 steps;
          run(mode="tiled")    accumulates it INSIDE the kernel's time loop (LDS-privatised, no scratch memory);
          run(mode="per_step") the per-step kernel writes the bin indices (or, hist_ring="T", T) of S steps into a ring strip and a
                               histogram launch counts the strip right behind it.
        concentration_driven: inverse mode — `emissions` holds the TARGET concentrations [n_steps, G]
        at the end of each step (shared by all members); the per-member emissions that reach them
        are diagnosed into `self.E` ([n_rows, G, N], aliasing `self.C`), and `self.cumE` [G, N] is
        extra per-member state.  Runs through `run()` as one time-fused launch per call.
        chunk_members: per-step / graph runs of ensembles whose state + parameters exceed the 256 MiB
        Infinity Cache are scheduled chunk-major — all requested steps for members [0, c), then
        [c, 2c), ... — so each chunk's rows stay cache-resident between its consecutive launches
        (+12-15 % at 4-8M members, bit-identical results; members never interact).  "auto" picks c
        from the bytes per member; an int forces it; None / 0 disables it.
        per_step_streams: mode='per_step' launches each timestep as this many kernels over contiguous member parts, each
        part's launches on its own HIP stream, so that one part's launch tail and ramp overlap the other part's kernel
        (members never interact, so nothing orders the parts against each other).  Two parts: -6.5 % per step at 1M fp64
        members (36.9 -> 34.5 us), -8 % at 0.5M, -2 % at 4M, +10 % at 0.25M (profiles/r03/two_stream_*.txt), bit-identical
        results.  "auto": 2 when one step moves at least ~16 us of traffic, else 1; an int forces it.
        fused_span: mode='fused' covers the requested steps with launches of this many steps (None: one launch).  A SIMD serves
        its oldest wave first, so the waves of a long launch finish in tiers and a launch with FEW rounds of waves ends in a
        long tail; relaunching the same kernel resets the ages (the state crosses HBM once per span: nothing at 128 steps).
        One launch against a relaunch every 128 steps, stats-only, fp64: -8 % at 0.1M members, -4..-5 % from 0.25M to 1.25M,
        -3.6 % at 2M, +0.7 % at 4M (profiles/r03/relaunch_sweep.txt).  "auto": FUSED_SPAN_STEPS when the ensemble is between
        FUSED_SPAN_MIN_ROUNDS and FUSED_SPAN_MAX_ROUNDS rounds of resident waves, else one launch.  Bit-identical either way."""
        if dtype not in _DTYPES:
            raise ValueError("dtype must be torch.float64 or torch.float32")
        self.lib = _capi.load(lib_path)    # raises if the HIP library is not built
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the ensemble engine has no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise ValueError(f"device {self.device}: the engine runs on a GPU only")
        self.dtype = dtype
        self._sfx = _DTYPES[dtype]
        self.n_members = N = int(n_members)
        if N < 1:
            raise ValueError("n_members must be >= 1")
        self.params = params
        self.n_gas = G = n_gas_of(params)
        self.pools = pools_of(params)
        self.sum_pools = SP = sum(self.pools)
        self.dt = float(dt)
        self.model = make_model(params, dt)
        n_pools = (ctypes.c_int32 * G)(*self.pools)
        if not self.lib.fiveeq_layout_supported(G, n_pools):
            raise _capi.FiveEqError(_capi.E_UNSUPPORTED, f"pool layout {self.pools} has no compiled kernel")

        if not store_trajectory:
            output_steps = []
        self.concentration_driven = bool(concentration_driven)
        drive = make_drive(emissions, F_ext, dt, output_steps, self.concentration_driven)
        self.out_steps = np.nonzero(drive[:, 7] >= 0)[0]          # step index of each stored row
        self.n_rows = int(self.out_steps.size)
        if drive[:, G:3].any():
            raise ValueError("emissions carry more gases than the parameter set")
        self.n_steps = int(drive.shape[0])

        dev, dt_ = self.device, dtype
        with torch.cuda.device(dev):
            self.drive = torch.from_numpy(drive).to(dev, dt_).contiguous()
            prm_rows = {k: _rows(params[k], G, N, k) for k in ("r0", "rC", "rT")}
            q_rows = _rows(params["q"], 2, N, "q")
            if any(isinstance(v, torch.Tensor) for v in list(prm_rows.values()) + [q_rows]):
                as_t = lambda v: v if isinstance(v, torch.Tensor) else torch.from_numpy(np.array(v, order="C"))  # noqa: E731
                self.r = torch.stack([as_t(prm_rows[k])[g].to(dev, dt_) for g in range(G) for k in ("r0", "rC", "rT")])
                self.q = as_t(q_rows).to(dev, dt_).contiguous()
            else:
                rows = np.concatenate([np.stack([prm_rows[k][g] for k in ("r0", "rC", "rT")]) for g in range(G)],
                                      axis=0)                                  # [3G, N]
                self.r = torch.from_numpy(np.array(rows, dtype=np.float64, order="C")).to(dev, dt_).contiguous()
                self.q = torch.from_numpy(np.array(q_rows, dtype=np.float64, order="C")).to(dev, dt_).contiguous()
            self.R = torch.zeros((SP, N), dtype=dt_, device=dev)
            self.S = torch.zeros((2, N), dtype=dt_, device=dev)
            # zero-filled, not torch.empty: rows of steps that were not run read as 0 rather than as stale
            # device memory (and the fill touches every page once, at construction)
            self.C = (torch.zeros((self.n_rows, G, N), dtype=dt_, device=dev)
                      if self.n_rows and (store_concentrations or concentration_driven) else None)
            self.T = torch.zeros((self.n_rows, N), dtype=dt_, device=dev) if self.n_rows else None
            self.cumE = torch.zeros((G, N), dtype=dt_, device=dev) if self.concentration_driven else None
            self.E = self.C if self.concentration_driven else None
            self.n_waves = int(self.lib.fiveeq_stats_waves(N))
            self.collect_stats = bool(collect_stats)
            self.T_stats = None          # per-wave records: 4.7 GB at 12.5M members x 750 steps, so allocated on demand
            # per-step (count, sum, sum^2, min, max) produced by the streamed histogram pass instead of the kernels
            self._step_sums = (torch.zeros((self.n_steps, 5), dtype=torch.float64, device=dev) if collect_stats else None)
            self._step_sums_valid = np.zeros(self.n_steps, dtype=bool)
            # steps whose moments THIS engine holds (wave records written by its launches, or records / folded sums restored
            # from a checkpoint): only these are saved as valid by state_dict("summaries")
            self._stats_have = np.zeros(self.n_steps, dtype=bool)
            self.hist_spec = None
            self.T_hist = None
            if hist is not None:
                lo_h, hi_h, nb = float(hist[0]), float(hist[1]), int(hist[2])
                if not (hi_h > lo_h) or not 1 <= nb <= 4096:
                    raise ValueError("hist=(lo, hi, n_bins): need lo < hi and 1 <= n_bins <= 4096")
                if self.concentration_driven:
                    raise ValueError("in-loop histograms are not available in concentration-driven mode")
                self.hist_spec = (lo_h, hi_h, nb)
                self.T_hist = torch.zeros((self.n_steps, nb), dtype=torch.int64, device=dev)
            if hist_ring_steps == "auto":      # as long as 8 GB of ring allow, at most 128 steps: every chunk boundary
                # costs one state + parameter round trip through HBM.  A ring entry is a 2-byte bin index (hist_ring='bins')
                # or a T value (hist_ring='T'): 12.5M fp32 members get 2 x 128 steps of bin indices (6.4 GB) where the T ring
                # stops at 2 x 85 — and the streamed histograms cost +18-20 % instead of +21-23 % (profiles/r04).
                w_ = 2 if hist_ring == "bins" else (8 if dtype == torch.float64 else 4)
                hist_ring_steps = min(128, max(8, (8 << 30) // (2 * N * w_)))
            self.hist_ring_steps = max(1, min(int(hist_ring_steps), self.n_steps))
            # where the streamed pipeline's histogram pass runs: "side" = a second HIP stream beside the next chunk's fused
            # kernel, "same" = behind each chunk on the caller's stream (see _run_fused_streamed_hist)
            self.hist_pass_stream = _env_choice("FIVEEQ_HIST_PASS_STREAM", "side", ("side", "same"))
            if hist_ring not in ("bins", "T"):
                raise ValueError("hist_ring must be 'bins' or 'T'")
            self.hist_ring = hist_ring
            self._ring = None            # allocated by the first streamed-histogram run
            self._bins = None
        if chunk_members == "auto":
            chunk_members = self.auto_chunk(N, SP, G, dtype)
        self.chunk_members = int(chunk_members or 0) // 256 * 256
        if per_step_streams == "auto":
            w_ = 8 if dtype == torch.float64 else 4
            t_step = min(N, self.chunk_members or N) * w_ * (2 * SP + 4 * G + 7) / HBM_STREAM_BYTES_PER_S
            per_step_streams = 2 if t_step >= PER_STEP_SPLIT_MIN_S else 1
        self.per_step_streams = max(1, int(per_step_streams))
        if fused_span not in ("auto", None) and int(fused_span) < 1:
            raise ValueError("fused_span must be 'auto', None or a positive number of steps")
        self.fused_span = fused_span if fused_span in ("auto", None) else int(fused_span)
        self.small_lanes = 0                # mode='small': lanes per member (0 = the widest form the layout has)
        self._ps_side = []                  # side streams of the per-step parts, created on first use
        self._ps_unjoined = False           # run(..., join=False) left work on the side streams the caller's has not waited for
        self._R0 = None if R0 is None else np.asarray(R0, dtype=np.float64).reshape(SP, N)
        self._S0 = None if S0 is None else np.asarray(S0, dtype=np.float64).reshape(2, N)
        self.t_next = 0                     # first step not yet run (bookkeeping for checkpoints)
        self.last_mode = None               # the mode the last run() used after resolving 'auto'
        self.reset_state()
        self._plans = {}

    @staticmethod
    def auto_chunk(n_members, sum_pools, n_gas, dtype, cache_bytes=INFINITY_CACHE_BYTES):
        """Members per chunk such that one chunk's state + parameter rows fill the Infinity Cache
        (rounded down to 65536 members); 0 = do not chunk (the ensemble is < 1.5 chunks)."""
        w = 8 if dtype == torch.float64 else 4
        resident = w * (sum_pools + 2 + 3 * n_gas + 2)
        c = (int(cache_bytes) // resident) // 65536 * 65536
        return c if n_members > c + c // 2 else 0

    def fused_span_steps(self, n_steps):
        """Steps per launch mode='fused' uses for a request of n_steps steps (see fused_span); n_steps = one launch."""
        n_steps = int(n_steps)
        if self.fused_span is None or n_steps <= 1:
            return n_steps
        if self.fused_span != "auto":
            return min(self.fused_span, n_steps)
        per_wave = 128 if self.dtype == torch.float32 and self.n_members % 2 == 0 else 64      # packed fp32 lanes carry two members
        slots = 16 * torch.cuda.get_device_properties(self.device).multi_processor_count       # 4 waves on each of a CU's 4 SIMDs
        rounds = -(-self.n_members // per_wave) / slots
        # (below a quarter of a round the launches themselves weigh more than the tail: 10k members lose 3 %)
        return min(FUSED_SPAN_STEPS, n_steps) if FUSED_SPAN_MIN_ROUNDS <= rounds <= FUSED_SPAN_MAX_ROUNDS else n_steps

    def auto_k_steps(self):
        """Steps per launch for mode='auto': 1 (the per-step kernel) while one step's HBM traffic hides the
        dependent-launch boundary, otherwise the K that brings a launch's traffic time to ~3 boundaries
        (capped at 32: a launch costs ~2.6 us beside ~0.75 us per step of a 10k-member ensemble — 0.92 us/step at K = 16,
        0.84 at 32, 0.75 fully fused, profiles/r03/in_loop_hist_config5_shard_f32.txt): small ensembles are launch-bound,
        not bandwidth-bound."""
        t_step = self.n_members * self.bytes_per_member_step("per_step") / HBM_STREAM_BYTES_PER_S
        if t_step >= 3.0 * LAUNCH_BOUNDARY_S:
            return 1
        return int(min(32, max(2, round(3.0 * LAUNCH_BOUNDARY_S / max(t_step, 1e-9)))))

    # -- state -------------------------------------------------------------------------
    def reset_state(self):
        """Back to the initial condition (zeros, or the R0/S0 given at construction) — the run accumulators too:
        `T_hist` ACCUMULATES over the runs that fill it (a step histogrammed twice counts twice), so it is zeroed here,
        and the per-step moment records are marked not-yet-written."""
        if self._ps_unjoined:
            self.join()
        if self._R0 is None:
            self.R.zero_()
        else:
            self.R.copy_(torch.from_numpy(self._R0).to(self.dtype))
        if self._S0 is None:
            self.S.zero_()
        else:
            self.S.copy_(torch.from_numpy(self._S0).to(self.dtype))
        if self.cumE is not None:
            self.cumE.zero_()
        if self.T_hist is not None:
            self.T_hist.zero_()
        self._step_sums_valid[:] = False
        self._stats_have[:] = False
        self.t_next = 0

    def state_dict(self, include_outputs="summaries"):
        """Checkpoint: everything a resumed run needs besides the (immutable) parameters and drive
        table — pools, thermal boxes, in inverse mode the per-member cumulative emissions, the index
        `t_next` of the first step not yet run — as host NumPy arrays (state in fp64), plus, by `include_outputs`:
          "summaries" (default)  what the run has REDUCED so far: `T_hist` ([n_steps, n_bins] int64, 24 MB at 750 x 4096)
                                 and the per-step moment sums of steps [0, t_next) folded to [n_steps, 5] (30 KB) — small
                                 whatever the ensemble size;
          True                   also the raw buffers: per-wave records `T_stats` (0.5 B per member-step: 4.7 GB at
                                 12.5M members x 750 steps) and the stored C/T rows ((G+1) w bytes per member and stored
                                 step: 24 GB for 1M fp64 members x 750 steps) — sized like the run, so opt-in;
          False                  the state only.
        Resume with `load_state_dict` and `run(state["t_next"], ...)`: bit-identical to an uninterrupted run
        (checkpoint/resume)."""
        if include_outputs not in (True, False, "summaries"):
            raise ValueError("include_outputs must be True, False or 'summaries'")
        torch.cuda.synchronize(self.device)
        out = {"R": self.R.double().cpu().numpy(), "S": self.S.double().cpu().numpy(), "t_next": int(self.t_next)}
        if self.cumE is not None:
            out["cumE"] = self.cumE.double().cpu().numpy()
        if include_outputs:
            if self.T_hist is not None:
                out["T_hist"] = self.T_hist.cpu().numpy()
            if self.collect_stats:
                # only the steps this engine HAS moments for (it ran them, or a checkpoint brought them): a run that began
                # at t_begin > 0, or a state-only checkpoint loaded before it, leaves the earlier steps out — their
                # zero-filled records are not moments
                sums = np.zeros((self.n_steps, 5), dtype=np.float64)
                valid = self._stats_have.copy()
                if valid.any():
                    lo_t, hi_t = int(np.nonzero(valid)[0][0]), int(np.nonzero(valid)[0][-1]) + 1
                    sums[lo_t:hi_t] = self.stats_sums(lo_t, hi_t).cpu().numpy()
                    sums[~valid] = 0.0
                out["_step_sums"], out["_step_sums_valid"] = sums, valid
        if include_outputs is True:
            for name in ("T_stats", "C", "T"):
                buf = getattr(self, name)
                if buf is not None:
                    out[name] = buf.cpu().numpy()
        return out

    def load_state_dict(self, state):
        """Restore a checkpoint (after `join()` if a run(..., join=False) is still outstanding).  One WITHOUT summaries (include_outputs=False) restores the state only: the accumulators
        of this engine (T_hist, per-step moments) are then cleared, because they describe a run this state is not from."""
        if self._ps_unjoined:
            self.join()
        for name in ("R", "S") + (("cumE",) if self.cumE is not None else ()):
            dst = getattr(self, name)
            src = np.asarray(state[name], dtype=np.float64)
            if src.shape != tuple(dst.shape):
                raise ValueError(f"{name}: checkpoint shape {src.shape}, engine {tuple(dst.shape)}")
            dst.copy_(torch.from_numpy(src).to(self.dtype))
        self.t_next = int(state.get("t_next", 0))
        self._step_sums_valid[:] = False
        self._stats_have[:] = False
        if "_step_sums_valid" in state and self._step_sums is not None:
            self._step_sums_valid[:] = np.asarray(state["_step_sums_valid"], dtype=bool)
            self._stats_have[:] = self._step_sums_valid
        if "T_stats" in state and self.collect_stats:          # raw wave records: every step before t_next was run by the saver
            self._stats_have[:self.t_next] = True
        if self.T_hist is not None and "T_hist" not in state:
            self.T_hist.zero_()
        if "T_stats" in state:
            self._wave_stats()                                   # the checkpoint carries wave records: make room for them
        for name in ("T_stats", "T_hist", "C", "T", "_step_sums"):
            dst = getattr(self, name)
            if dst is not None and name in state:
                src = np.asarray(state[name])
                if src.shape != tuple(dst.shape):
                    raise ValueError(f"{name}: checkpoint shape {src.shape}, engine {tuple(dst.shape)}")
                dst.copy_(torch.from_numpy(src).to(dst.dtype))

    def _wave_stats(self):
        """The per-wave record buffer of the in-kernel statistics, allocated by the first launch that writes it (the
        streamed histogram pipeline takes its moments from the histogram pass and never needs it)."""
        if self.collect_stats and self.T_stats is None:
            self.T_stats = torch.zeros((self.n_waves, self.n_steps, 4), dtype=torch.float64, device=self.device)
        return self.T_stats

    # -- launches ----------------------------------------------------------------------
    def _stream(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        return ctypes.c_void_p(s.cuda_stream)

    def _ptr(self, t):
        return ctypes.c_void_p(0 if t is None else t.data_ptr())

    def _run_args(self, t_begin, t_end, m0=0, n=None):
        """C-ABI arguments for members [m0, m0 + n) of this engine's rows (ld = N)."""
        N = self.n_members
        n = N if n is None else n
        w = 8 if self.dtype == torch.float64 else 4

        def at(t, byte_off):
            return ctypes.c_void_p(0 if t is None else t.data_ptr() + byte_off)

        return (ctypes.byref(self.model), n, N, self._ptr(self.drive), self.n_steps, int(t_begin), int(t_end),
                at(self.r, m0 * w), at(self.q, m0 * w), at(self.R, m0 * w), at(self.S, m0 * w),
                at(self.C, m0 * w), at(self.T, m0 * w), self.n_rows,
                at(self.T_stats, (m0 // 64) * self.n_steps * 4 * 8))

    def _chunks(self):
        N, c = self.n_members, self.chunk_members
        if not c or c >= N:
            return [(0, N)]
        return [(m0, min(c, N - m0)) for m0 in range(0, N, c)]

    def _run_inverse(self, t_begin, t_end, stream):
        N = self.n_members
        fn = getattr(self.lib, f"fiveeq_run_inverse_{self._sfx}")
        return fn(ctypes.byref(self.model), N, N, self._ptr(self.drive), self.n_steps, int(t_begin), int(t_end),
                  self._ptr(self.r), self._ptr(self.q), self._ptr(self.R), self._ptr(self.S), self._ptr(self.cumE),
                  self._ptr(self.C), self._ptr(self.T), self.n_rows, self._ptr(self.T_stats), self._stream(stream))

    def step(self, t, stream=None):
        """One timestep = one kernel launch (asynchronous)."""
        if self._ps_unjoined:                         # a run(..., join=False) may still be writing R, S, T_stats on the side streams
            self.join(stream)
        self._step_sums_valid[int(t)] = False         # this launch writes the step's wave record: older folded moments are stale
        if self.collect_stats:
            self._stats_have[int(t)] = True
        if self.concentration_driven:
            self._wave_stats()
            with torch.cuda.device(self.device):
                _capi.check(self.lib, self._run_inverse(t, t + 1, stream))
            self.t_next = int(t) + 1
            return
        N = self.n_members
        fn = getattr(self.lib, f"fiveeq_step_{self._sfx}")
        self._wave_stats()
        with torch.cuda.device(self.device):
            rc = fn(ctypes.byref(self.model), N, N, self._ptr(self.drive), self.n_steps, int(t),
                    self._ptr(self.r), self._ptr(self.q), self._ptr(self.R), self._ptr(self.S),
                    self._ptr(self.C), self._ptr(self.T), self.n_rows, self._ptr(self.T_stats),
                    self._stream(stream))
        _capi.check(self.lib, rc)
        self.t_next = int(t) + 1

    def run(self, t_begin=0, t_end=None, mode="per_step", stream=None, k_steps=None, join=True):
        """Advance steps [t_begin, t_end).  mode:
        'per_step' one launch per timestep, enqueued from C (the north-star form);
        'graph'    the same launches replayed from a captured hipGraph;
        'fused'    one launch, state in registers across all steps (with `hist=`: chunks of hist_ring_steps steps, T_hist
                   filled by the streamed pipeline, see __init__);
        'ksteps'   the fused kernel over consecutive spans of `k_steps` steps (default `auto_k_steps()`):
                   state crosses HBM once per k_steps — the per-step family's answer for small ensembles;
        'tiled'    the time-tiled persistent kernel, `k_steps` steps per launch (None/0: the largest tile
                   that fits the LDS); accumulates `T_hist` inside the time loop if the engine has `hist=`;
        'auto'     'per_step' while a step's HBM traffic hides the launch boundary, else 'ksteps' — or, on an engine with
                   hist=, 'fused' (the streamed histogram pipeline), the fastest form that fills T_hist on a launch-bound
                   ensemble (profiles/r04/auto_hist_table.txt).
        Every mode gives bit-identical results.
        join=False (mode 'per_step' on several streams only): do not make the caller's stream wait for the side streams at
        the end, and do not make the side streams wait for the caller's stream at the start of the NEXT such call — for
        back-to-back calls with nothing in between that touches the state on the caller's stream (bench.py's repeated
        blocks): a join is two cross-stream hops, ~20 us.  Call `join()` before anything consumes the results."""
        t_end = self.n_steps if t_end is None else int(t_end)
        if self._ps_unjoined and mode != "per_step":
            self.join(stream)
        if mode == "auto":
            k_steps = self.auto_k_steps() if k_steps is None else int(k_steps)
            if k_steps <= 1:
                mode = "per_step"
            elif self.T_hist is not None:
                # Launch-bound AND filling T_hist: the streamed pipeline (the fused kernel in chunks of hist_ring_steps steps
                # + the histogram pass) — measured against the other two forms that fill T_hist, us per step at 10k / 100k
                # members, 4096 bins, fp64: 1.6 / 2.5 against per-step + bins 4.2 / 6.4 and the tiled kernel at the auto K
                # 4.4 / 10.6 (its persistent grid and per-launch flush need long tiles and millions of members); same table
                # for fp32 and 1024 bins: profiles/r04/auto_hist_table.txt, DESIGN.md section 3.5.  Rounds 2-3 sent these
                # ensembles to the tiled kernel.
                mode, k_steps = "fused", None
            else:
                mode = "ksteps"
        self.last_mode = mode            # what 'auto' resolved to (tests, bench.py's config.mode_resolved)
        if self.T_hist is not None and mode in ("graph", "ksteps"):
            raise ValueError(f"mode {mode!r} does not fill T_hist: use 'fused', 'tiled' or 'per_step' with hist=")
        with torch.cuda.device(self.device):
            if not (mode == "fused" and self.T_hist is not None and not self.concentration_driven and self.hist_ring == "T"):
                self._wave_stats()
                # these launches write per-wave records: moments an earlier streamed pass left for the same steps are stale
                self._step_sums_valid[int(t_begin):t_end] = False
            if self.collect_stats:
                self._stats_have[int(t_begin):t_end] = True
            if self.concentration_driven:
                rc = self._run_inverse(t_begin, t_end, stream)
            elif mode == "per_step" and self.T_hist is not None:
                rc = self._run_per_step_hist(t_begin, t_end, stream)
            elif mode == "per_step":
                rc = self._run_per_step(t_begin, t_end, stream, join)
            elif mode == "fused" and self.T_hist is not None and self.hist_ring == "bins":
                rc = self._run_fused_bin_ring(t_begin, t_end, stream)
            elif mode == "fused" and self.T_hist is not None:
                rc = self._run_fused_streamed_hist(t_begin, t_end, stream)
            elif mode == "fused":
                span = self.fused_span_steps(t_end - int(t_begin))
                if span < t_end - int(t_begin):                  # the same kernel, relaunched every `span` steps
                    fn = getattr(self.lib, f"fiveeq_run_ksteps_{self._sfx}")
                    rc = fn(*self._run_args(t_begin, t_end), span, self._stream(stream))
                else:
                    fn = getattr(self.lib, f"fiveeq_run_fused_{self._sfx}")
                    rc = fn(*self._run_args(t_begin, t_end), self._stream(stream))
            elif mode == "ksteps":
                k = self.auto_k_steps() if k_steps is None else int(k_steps)
                fn = getattr(self.lib, f"fiveeq_run_ksteps_{self._sfx}")
                rc = fn(*self._run_args(t_begin, t_end), max(k, 1), self._stream(stream))
            elif mode == "small":
                fn = getattr(self.lib, f"fiveeq_run_small_{self._sfx}")
                rc = fn(*self._run_args(t_begin, t_end)[:-1], int(self.small_lanes), self._stream(stream))
            elif mode == "tiled":
                fn = getattr(self.lib, f"fiveeq_run_tiled_{self._sfx}")
                lo_h, hi_h, nb = self.hist_spec if self.hist_spec is not None else (0.0, 1.0, 0)
                rc = fn(*self._run_args(t_begin, t_end), int(k_steps or 0), lo_h, hi_h, nb, self._ptr(self.T_hist),
                        self._stream(stream))
            elif mode == "graph":
                # one captured plan per (chunk, part), the parts of a chunk replayed side by side on their own streams
                rc = _capi.OK
                plans = self.prepare_graph(t_begin, t_end)
                streams = self.per_step_stream_list(stream)
                if self._ps_unjoined:
                    self.join(stream)
                for s_ in streams[1:]:
                    s_.wait_stream(streams[0])
                for plan, (_, _, si) in zip(plans, self.per_step_launches()):
                    rc = rc or self.lib.fiveeq_plan_launch(plan, self._stream(streams[si]))
                for s_ in streams[1:]:
                    streams[0].wait_stream(s_)
            else:
                raise ValueError(f"unknown mode {mode!r}")
        _capi.check(self.lib, rc)
        self.t_next = t_end

    def per_step_launches(self):
        """[(first member, members, stream index)] of the kernels mode='per_step' launches for ONE timestep, in launch order:
        member chunks run one after the other (chunk-major, see chunk_members), the parts of a chunk side by side on
        per_step_streams streams (stream 0 = the caller's)."""
        out = []
        for m0, n in self._chunks():
            k = self.per_step_streams if n >= 512 * self.per_step_streams else 1
            cuts = [m0 + (i * n // k) // 256 * 256 for i in range(k)] + [m0 + n]
            out += [(cuts[i], cuts[i + 1] - cuts[i], i) for i in range(k)]
        return out

    def per_step_stream_list(self, stream=None):
        """The HIP streams mode='per_step' launches on: the caller's, then the side streams of the other parts."""
        main = stream if stream is not None else torch.cuda.current_stream(self.device)
        n_streams = 1 + max(i for _, _, i in self.per_step_launches())
        while len(self._ps_side) < n_streams - 1:
            self._ps_side.append(torch.cuda.Stream(device=self.device))
        return [main] + self._ps_side[:n_streams - 1]

    def join(self, stream=None):
        """Make the caller's stream wait for everything run(..., join=False) enqueued on the side streams."""
        streams = self.per_step_stream_list(stream)
        for s in streams[1:]:
            streams[0].wait_stream(s)
        self._ps_unjoined = False

    def _run_per_step(self, t_begin, t_end, stream, join=True):
        fn = getattr(self.lib, f"fiveeq_run_{self._sfx}")
        launches = self.per_step_launches()
        streams = self.per_step_stream_list(stream)
        main = streams[0]
        if len(streams) == 1:
            rc = _capi.OK
            for m0, n, _ in launches:                             # chunk-major: see chunk_members
                rc = rc or fn(*self._run_args(t_begin, t_end, m0, n), self._stream(main))
            return rc
        if not self._ps_unjoined:
            for s in streams[1:]:
                s.wait_stream(main)                               # the state may have been touched on the caller's stream
        rc = _capi.OK
        chunk_first = [i for i, (_, _, si) in enumerate(launches) if si == 0]
        for ci, first in enumerate(chunk_first):                  # chunks one after the other, their parts side by side
            group = launches[first:chunk_first[ci + 1] if ci + 1 < len(chunk_first) else len(launches)]
            for t in range(int(t_begin), int(t_end), PER_STEP_BLOCK):     # short blocks keep every stream's queue fed
                t1 = min(int(t_end), t + PER_STEP_BLOCK)
                for m0, n, si in group:
                    rc = rc or fn(*self._run_args(t, t1, m0, n), self._stream(streams[si]))
        if join:
            for s in streams[1:]:
                main.wait_stream(s)
        self._ps_unjoined = not join
        return rc

    def _hist_ring(self, slots=2):
        """Ring [slots, S, N] of T rows + the drive table whose output row is t mod S (shared by the streamed pipelines:
        mode 'fused' double-buffers, two slots; mode 'per_step' uses one)."""
        N, S, dev = self.n_members, max(1, min(int(self.hist_ring_steps), self.n_steps)), self.device
        if self._ring is None or self._ring["S"] != S or self._ring["buf"].shape[0] < slots:
            torch.cuda.synchronize(dev)      # (re)built whenever hist_ring_steps changed: the kernels are told n_rows = S
            self._ring = None                # and must find S rows behind the pointer
            drive = self.drive.clone()
            drive[:, 7] = torch.arange(self.n_steps, device=dev, dtype=torch.int64).remainder(S).to(self.dtype)
            self._ring = {"S": S, "drive": drive, "buf": torch.empty((slots, S, N), dtype=self.dtype, device=dev),
                          "side": torch.cuda.Stream(device=dev), "drained": [torch.cuda.Event(), torch.cuda.Event()]}
        return self._ring

    def _run_per_step_hist(self, t_begin, t_end, stream):
        """mode='per_step' with hist=: the per-step kernel (one launch per timestep, enqueued from C, chunk-major like
        the plain per-step path) stores T of S = hist_ring_steps consecutive steps into a ring strip, then ONE histogram
        launch over those S rows adds them into T_hist[t:t+S].  Two C calls per S steps and member chunk; the in-kernel
        per-wave moments stay on (they are free in this kernel)."""
        if self.hist_ring == "bins":
            return self._run_per_step_bin_ring(t_begin, t_end, stream)
        if self.C is not None:
            raise RuntimeError("per-step histograms through a T ring carry T only: build the engine with "
                               "store_concentrations=False (or store_trajectory=False), or use hist_ring='bins'")
        N = self.n_members
        ring = self._hist_ring(slots=1)
        S = ring["S"]
        buf = ring["buf"][0]
        run = getattr(self.lib, f"fiveeq_run_{self._sfx}")
        hist = getattr(self.lib, f"fiveeq_hist_rows_{self._sfx}")
        lo_h, hi_h, nb = self.hist_spec
        stored = {int(t): r for r, t in enumerate(self.out_steps)}
        w = 8 if self.dtype == torch.float64 else 4
        at = lambda t, off: ctypes.c_void_p(0 if t is None else t.data_ptr() + off)   # noqa: E731
        # the same launch layout as the plain per-step path: chunks one after the other, the parts of a chunk side by
        # side on their own streams; every part runs its S steps and then histograms its own strip of the ring
        launches = self.per_step_launches()
        streams = self.per_step_stream_list(stream)
        if self._ps_unjoined:
            self.join(stream)
        for s_ in streams[1:]:
            s_.wait_stream(streams[0])
        chunk_first = [i for i, (_, _, si) in enumerate(launches) if si == 0]
        rc = _capi.OK
        for ci, first in enumerate(chunk_first):
            group = launches[first:chunk_first[ci + 1] if ci + 1 < len(chunk_first) else len(launches)]
            t = int(t_begin)
            while t < t_end and rc == _capi.OK:
                t1 = min(int(t_end), (t // S + 1) * S)
                for m0, n, si in group:
                    st = self._stream(streams[si])
                    rc = rc or run(ctypes.byref(self.model), n, N, self._ptr(ring["drive"]), self.n_steps, t, t1,
                                   at(self.r, m0 * w), at(self.q, m0 * w), at(self.R, m0 * w), at(self.S, m0 * w),
                                   ctypes.c_void_p(0), at(buf, m0 * w), S,
                                   at(self.T_stats, (m0 // 64) * self.n_steps * 4 * 8), st)
                    rc = rc or hist(t1 - t, n, N, at(buf[t % S], m0 * w), lo_h, hi_h, nb, self._ptr(self.T_hist[t:t1]), st)
                    with torch.cuda.stream(streams[si]):
                        for tt in range(t, t1):
                            if tt in stored:
                                self.T[stored[tt], m0:m0 + n].copy_(buf[tt % S, m0:m0 + n])
                t = t1
        for s_ in streams[1:]:
            streams[0].wait_stream(s_)
        return rc

    def _run_per_step_bin_ring(self, t_begin, t_end, stream):
        """mode='per_step' with hist= and hist_ring='bins': fiveeq_run_bins_* — the per-step kernel, one launch per timestep
        and member part, the engine's own drive table, stored C/T rows and wave records as in a plain per-step run — also
        writes every member's histogram bin into a ring strip [S, N] of uint16 (row t mod S); after S steps each part counts
        its strip into T_hist (fiveeq_hist_bins) on its own stream.  2 bytes written + 2 read per member-step on top of the
        step's 124 / 248."""
        N = self.n_members
        ring = self._bin_ring(slots=1)
        S = ring["S"]
        buf = ring["buf"][0]
        run = getattr(self.lib, f"fiveeq_run_bins_{self._sfx}")
        lo_h, hi_h, nb = self.hist_spec
        at = lambda t, off: ctypes.c_void_p(0 if t is None else t.data_ptr() + off)   # noqa: E731
        launches = self.per_step_launches()
        streams = self.per_step_stream_list(stream)
        if self._ps_unjoined:
            self.join(stream)
        for s_ in streams[1:]:
            s_.wait_stream(streams[0])
        chunk_first = [i for i, (_, _, si) in enumerate(launches) if si == 0]
        rc = _capi.OK
        for ci, first in enumerate(chunk_first):
            group = launches[first:chunk_first[ci + 1] if ci + 1 < len(chunk_first) else len(launches)]
            t = int(t_begin)
            while t < t_end and rc == _capi.OK:
                t1 = min(int(t_end), (t // S + 1) * S)
                for m0, n, si in group:
                    st = self._stream(streams[si])
                    rc = rc or run(*self._run_args(t, t1, m0, n), lo_h, hi_h, nb, at(buf, m0 * 2), S, st)
                    rc = rc or self.lib.fiveeq_hist_bins(t1 - t, n, N, at(buf[t % S], m0 * 2), nb,
                                                         self._ptr(self.T_hist[t:t1]), st)
                t = t1
        for s_ in streams[1:]:
            streams[0].wait_stream(s_)
        return rc

    def _bin_ring(self, slots=2):
        """Ring [slots, S, N] of uint16 bin indices for the streamed histograms (hist_ring='bins'): mode 'fused'
        double-buffers (two slots), mode 'per_step' counts each strip right behind its steps (one slot: half the memory,
        1.6 GB instead of 3.2 at 12.5M members and S = 64); grown to two slots when a fused run follows a per-step one."""
        N, S, dev = self.n_members, max(1, min(int(self.hist_ring_steps), self.n_steps)), self.device
        ring = self._bins
        if ring is None or ring["S"] != S or ring["buf"].shape[0] < slots:
            torch.cuda.synchronize(dev)
            self._bins = None
            self._bins = ring = {"S": S, "buf": torch.empty((slots, S, N), dtype=torch.int16, device=dev),
                                 "side": torch.cuda.Stream(device=dev), "drained": [torch.cuda.Event(), torch.cuda.Event()]}
        return ring

    def _run_fused_bin_ring(self, t_begin, t_end, stream):
        """mode='fused' with hist= and hist_ring='bins': chunks of S = hist_ring_steps steps.  Stream A (the caller's) runs
        fiveeq_run_fused_bins_* for chunk i: the ordinary fused kernel — the engine's own drive table, stored C/T rows and
        per-wave statistics exactly as in a plain fused run — that also writes every member's histogram bin of every step
        into ring slot i % 2 (row t mod S, one uint16 per member).  Stream B waits for the chunk and counts the slot's rows into
        T_hist[t:t+S] (fiveeq_hist_bins); A reuses a slot only after B has drained it.  Half (fp32) or a quarter (fp64) of
        the T ring's traffic in both directions; the moments come from the kernel's wave records."""
        N, dev = self.n_members, self.device
        ring = self._bin_ring()
        S = ring["S"]
        main = stream if stream is not None else torch.cuda.current_stream(dev)
        side = main if self.hist_pass_stream == "same" else ring["side"]
        side.wait_stream(main)
        fused = getattr(self.lib, f"fiveeq_run_fused_bins_{self._sfx}")
        lo_h, hi_h, nb = self.hist_spec
        used = [False, False]
        rc, t, i = _capi.OK, int(t_begin), 0
        while t < t_end and rc == _capi.OK:
            t1 = min(t_end, (t // S + 1) * S)              # chunks end on multiples of S: row = t mod S never wraps
            slot = i % 2
            buf = ring["buf"][slot]
            if used[slot]:
                main.wait_event(ring["drained"][slot])
            rc = fused(*self._run_args(t, t1), lo_h, hi_h, nb, self._ptr(buf), S, ctypes.c_void_p(main.cuda_stream))
            side.wait_stream(main)
            if rc == _capi.OK:
                with torch.cuda.stream(side):
                    rc = self.lib.fiveeq_hist_bins(t1 - t, N, N, self._ptr(buf[t % S:]), nb, self._ptr(self.T_hist[t:t1]),
                                                   ctypes.c_void_p(side.cuda_stream))
                ring["drained"][slot].record(side)
                used[slot] = True
            t, i = t1, i + 1
        main.wait_stream(side)
        return rc

    def _run_fused_streamed_hist(self, t_begin, t_end, stream):
        """mode='fused' with hist=: chunks of S = hist_ring_steps steps.  Stream A (the caller's) runs the fused kernel
        for chunk i with T of every step stored into ring slot i % 2 (its own drive table: output row = t mod S);
        stream B waits for that chunk, histograms the slot's rows into T_hist[t:t+S] (fiveeq_hist_rows_*) and copies
        the rows of the engine's own stored years into self.T; stream A reuses a slot only after B has drained it.
        The pass also returns the rows' moments, so the fused kernel runs without in-kernel statistics.  (Measured: the
        two streams do not hide the pass — it takes wave slots from the fused kernel — the pipeline costs the sum of its
        parts, +18 % at 12.5M fp32 members)"""
        if self.C is not None:
            raise RuntimeError("streamed histograms carry T only: build the engine with store_concentrations=False "
                               "(or store_trajectory=False), or use mode='tiled'")
        N = self.n_members
        dev = self.device
        ring = self._hist_ring()
        S = ring["S"]
        main = stream if stream is not None else torch.cuda.current_stream(dev)
        side = main if self.hist_pass_stream == "same" else ring["side"]
        side.wait_stream(main)                   # T_hist / self.T may have been touched on the caller's stream
        fused = getattr(self.lib, f"fiveeq_run_fused_{self._sfx}")
        with_stats = self._step_sums is not None
        hist = getattr(self.lib, f"fiveeq_hist_rows_{'stats_' if with_stats else ''}{self._sfx}")
        lo_h, hi_h, nb = self.hist_spec
        stored = {int(t): row for row, t in enumerate(self.out_steps)}
        used = [False, False]
        rc = _capi.OK
        t = int(t_begin)
        i = 0
        while t < t_end and rc == _capi.OK:
            t1 = min(t_end, (t // S + 1) * S)              # chunks end on multiples of S: row = t mod S never wraps
            slot = i % 2
            buf = ring["buf"][slot]
            if used[slot]:
                main.wait_event(ring["drained"][slot])
            # no in-kernel statistics here: the histogram pass reads every T anyway and returns the moments with it
            rc = fused(ctypes.byref(self.model), N, N, self._ptr(ring["drive"]), self.n_steps, t, t1, self._ptr(self.r),
                       self._ptr(self.q), self._ptr(self.R), self._ptr(self.S), ctypes.c_void_p(0), self._ptr(buf), S,
                       ctypes.c_void_p(0), ctypes.c_void_p(main.cuda_stream))
            side.wait_stream(main)
            if rc == _capi.OK:
                r0, k = t % S, t1 - t
                rows = buf[r0:r0 + k]
                with torch.cuda.stream(side):
                    if with_stats:
                        n_ch = int(self.lib.fiveeq_hist_rows_chunks(k, N))
                        mom = torch.empty((k, n_ch, 4), dtype=torch.float64, device=dev)
                        rc = hist(k, N, N, self._ptr(rows), lo_h, hi_h, nb, self._ptr(self.T_hist[t:t1]), self._ptr(mom),
                                  ctypes.c_void_p(side.cuda_stream))
                        sums = self._step_sums[t:t1]
                        sums[:, 0] = float(N)
                        sums[:, 1:3] = mom[:, :, 0:2].sum(dim=1)
                        sums[:, 3] = mom[:, :, 2].min(dim=1).values
                        sums[:, 4] = mom[:, :, 3].max(dim=1).values
                        self._step_sums_valid[t:t1] = True
                    else:
                        rc = hist(k, N, N, self._ptr(rows), lo_h, hi_h, nb, self._ptr(self.T_hist[t:t1]),
                                  ctypes.c_void_p(side.cuda_stream))
                    for tt in range(t, t1):
                        if tt in stored:
                            self.T[stored[tt]].copy_(buf[tt % S])
                ring["drained"][slot].record(side)
                used[slot] = True
            t = t1
            i += 1
        main.wait_stream(side)
        return rc

    def tile_steps(self):
        """Steps per launch of mode='tiled' when k_steps is left to the library (LDS budget)."""
        nb = self.hist_spec[2] if self.hist_spec is not None else 0
        return int(getattr(self.lib, f"fiveeq_tile_steps_{self._sfx}")(nb))

    def prepare_graph(self, t_begin=0, t_end=None):
        """Capture (once) the per-step launches of [t_begin, t_end) into hipGraph plans, one per launch of
        `per_step_launches()` (member chunk x part); returns the list of plans in that order."""
        t_end = self.n_steps if t_end is None else int(t_end)
        key = (int(t_begin), t_end)
        plans = self._plans.get(key)
        if plans is None:
            plans = []
            self._wave_stats()
            fn = getattr(self.lib, f"fiveeq_plan_create_{self._sfx}")
            with torch.cuda.device(self.device):
                for m0, n, _ in self.per_step_launches():
                    plan = ctypes.c_void_p()
                    _capi.check(self.lib, fn(*self._run_args(t_begin, t_end, m0, n), ctypes.byref(plan)))
                    plans.append(plan)
            self._plans[key] = plans
        return plans

    def close(self):
        for plans in self._plans.values():
            for plan in plans:
                self.lib.fiveeq_plan_destroy(plan)
        self._plans = {}

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    # -- on-device summary statistics -----------------------------------------------------
    def stats_sums(self, t_begin=0, t_end=None):
        """[n, 5] fp64 per step: (count, sum T, sum T^2, min T, max T) over this shard's members — folded over the
        per-wave records the kernels wrote, or, for steps that ran through the streamed histogram pipeline
        (mode='fused' with hist=), as returned by that pass.  Additive across shards
        (fiveeqscm_amd.distributed.reduce_stats)."""
        if not self.collect_stats:
            raise RuntimeError("engine was built with collect_stats=False")
        if self._ps_unjoined:
            self.join()
        t_end = self.n_steps if t_end is None else int(t_end)
        valid = self._step_sums_valid[t_begin:t_end]
        if valid.all():                                          # everything came from the histogram pass
            return self._step_sums[t_begin:t_end].clone()
        s = self._wave_stats()[:, t_begin:t_end]                 # [W, n, 4]
        cnt = torch.full((s.shape[1],), float(self.n_members), dtype=torch.float64, device=s.device)
        out = torch.stack([cnt, s[:, :, 0].sum(0), s[:, :, 1].sum(0), s[:, :, 2].min(0).values,
                           s[:, :, 3].max(0).values], dim=1)
        if valid.any():
            pick = torch.from_numpy(valid).to(out.device)
            out[pick] = self._step_sums[t_begin:t_end][pick]
        return out

    def stats(self, t_begin=0, t_end=None):
        """dict of per-step ensemble moments of T over this shard: mean, var (population), min, max."""
        from .distributed import moments_from_sums
        return moments_from_sums(self.stats_sums(t_begin, t_end))

    def gather_summary(self, steps, percentiles=(5.0, 50.0, 95.0), dst=0, group=None, stats=None):
        """End-of-run summary of T at the stored `steps` over ALL members of all ranks (collective over `group`; see
        distributed.gather_summary): merged moments on every rank, exact percentiles on rank `dst`.  With
        collect_stats the moments come from the records the kernels wrote while stepping — the summary then reads the
        rows twice (histogram, selection) instead of three times."""
        from .distributed import gather_summary
        if self.T is None:
            raise RuntimeError("no stored T rows to summarise")
        if self._ps_unjoined:
            self.join()
        row_of = {int(t): r for r, t in enumerate(self.out_steps)}
        missing = [int(t) for t in steps if int(t) not in row_of]
        if missing:
            raise ValueError(f"steps {missing} are not stored (out_steps)")
        rows = self.T[[row_of[int(t)] for t in steps]]
        sums = None
        if self.collect_stats and all(self._stats_have[int(t)] for t in steps):       # else: the moments pass over the rows
            sums = torch.cat([self.stats_sums(int(t), int(t) + 1) for t in steps])[:, 1:5].contiguous()
        return gather_summary(rows, percentiles, dst=dst, group=group, stats=stats, local_sums=sums)

    def T_histogram(self, lo, hi, n_bins=4096, rows=None, out=None, stream=None):
        """Fixed-bin histograms of the stored T rows on the device: int64 tensor [n_rows, n_bins]
        (bin b counts lo + b w <= T < lo + (b+1) w; outliers land in the edge bins).  `out` lets
        several calls / shards accumulate; percentiles: distributed.histogram_percentiles."""
        if self.T is None:
            raise RuntimeError("no stored T rows to histogram")
        if self._ps_unjoined:
            self.join(stream)
        x = self.T if rows is None else self.T[rows].contiguous()
        k = x.shape[0]
        if out is None:
            out = torch.zeros((k, int(n_bins)), dtype=torch.int64, device=self.device)
        fn = getattr(self.lib, f"fiveeq_hist_rows_{self._sfx}")
        with torch.cuda.device(self.device):
            rc = fn(k, self.n_members, x.shape[1], self._ptr(x), float(lo), float(hi), int(n_bins), self._ptr(out),
                    self._stream(stream))
        _capi.check(self.lib, rc)
        return out

    def hist_edge_counts(self):
        """[n_steps, 2] int64: members counted in the first / last bin of T_hist per step.  Values outside [lo, hi) land
        there, so non-zero entries at steps whose true extremes lie inside the range mean the range was too tight for
        percentiles near that tail (compare with stats()['min'/'max'])."""
        if self.T_hist is None:
            raise RuntimeError("engine was built without hist=")
        if self._ps_unjoined:
            self.join()
        return torch.stack([self.T_hist[:, 0], self.T_hist[:, -1]], dim=1)

    # -- accounting ----------------------------------------------------------------------
    def bytes_per_member_step(self, mode="per_step", k_steps=None):
        """ALGORITHMIC HBM bytes per member-timestep:
        per_step:        w (2 SP + 4 G + 7)   [R,S read+write; r,q read; C,T write];
        fused:           w (G + 1) + w (2 SP + 3 G + 6) / steps per launch (n_steps, or fused_span);
        ksteps / tiled:  w (G + 1) + w (2 SP + 3 G + 6) / k_steps  (state + parameters once per k_steps).
        With `hist=` the streamed pipelines of 'fused' and 'per_step' add the ring: T written by the step kernel and read
        back by the histogram pass, 2 w per member-step, and 'fused' then reloads state + parameters once per
        hist_ring_steps instead of once per run (its moments come from the pass: no wave records)."""
        w = 8 if self.dtype == torch.float64 else 4
        G, SP = self.n_gas, self.sum_pools
        out = ((G if self.C is not None else 0) + 1) * self.n_rows / self.n_steps      # stored rows only
        extra = (32.0 / 64.0) if self.collect_stats else 0.0          # one 32-B stats record per wave
        ring = 2.0 * w if (self.T_hist is not None and mode in ("fused", "per_step")) else 0.0
        if mode == "fused" and self.T_hist is not None:
            if self.hist_ring == "bins":                       # 2 B written + 2 B read per member-step; wave records stay
                ring = 4.0 + extra
            return w * (out + (2 * SP + 3 * G + 6) / max(1, min(self.hist_ring_steps, self.n_steps))) + ring
        if mode == "fused":                                     # state + parameters once per launch of fused_span_steps() steps
            return w * (out + (2 * SP + 3 * G + 6) / self.fused_span_steps(self.n_steps)) + extra
        if mode in ("ksteps", "tiled"):
            k = k_steps or (self.auto_k_steps() if mode == "ksteps" else self.tile_steps())
            return w * (out + (2 * SP + 3 * G + 6) / max(int(k), 1)) + extra
        if mode == "per_step" and self.T_hist is not None and self.hist_ring == "bins":
            ring = 4.0
        return w * (2 * SP + 3 * G + 6 + out) + extra + ring


def run_ensemble(emissions, params, n_members, *, F_ext=None, dt=1.0, dtype=torch.float64, device=None,
                 mode="per_step", output_steps=None, collect_stats=False, R0=None, S0=None):
    """Whole-series convenience wrapper: returns dict(C [n_rows,G,N], T [n_rows,N], R, S, out_steps
    [, T_stats]) of device tensors after a synchronise."""
    eng = EnsembleEngine(params, n_members, emissions, F_ext=F_ext, dt=dt, dtype=dtype, device=device,
                         output_steps=output_steps, collect_stats=collect_stats, R0=R0, S0=S0)
    eng.run(mode=mode)
    torch.cuda.synchronize(eng.device)
    out = {"C": eng.C, "T": eng.T, "R": eng.R, "S": eng.S, "out_steps": eng.out_steps}
    if collect_stats:
        out["T_stats"] = eng.stats()
    eng.close()
    return out
