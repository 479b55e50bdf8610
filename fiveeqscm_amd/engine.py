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
    T_hist [n_steps, n_bins] optional fixed-bin histogram of T for EVERY step (int64), filled through a ring of 2-byte bin
                             indices that a histogram pass drains while the next steps are computed

There is no CPU path: constructing an engine without a GPU, or without the built
HIP library, raises.  (The reference's own function, `calculate_hfc_conc`, is a
NumPy one-liner and lives in fiveeqscm_amd.concentrations.)
"""
import ctypes
import os

import numpy as np
import torch

from . import _capi
from .checkpoint import CheckpointMixin
from .emissions import make_drive
from .params import make_model, n_gas_of, pools_of
from .tuning import (_env_choice, _env_positive, calibrate, concurrent_side_streams,  # noqa: F401  (calibrate: part of this
                     side_stream_report)                                               # module's interface)

_DTYPES = {torch.float64: "f64", torch.float32: "f32"}
INFINITY_CACHE_BYTES = 256 << 20     # MI355X die-level L3 (MI355X_MICROARCH.md); sizes the chunk-major schedule
# ... whose chunks hold at most this share of it in state + parameter rows: the per-step rate of a chunk-major run is flat from
# 0.55 to 0.85 of the cache and falls off above (8M fp64 members, two streams: 0.88-0.89 of 8 TB/s up to 0.8, 0.82-0.84 at
# 0.9-0.95; 25M fp32: 0.905-0.914 up to 0.85, 0.844 at 0.95; profiles/r05/chunk_share_sweep.txt; rounds 2-4 filled the whole
# cache, 0.965 after rounding).  An ensemble that fits the cache by itself is not chunked (0.91 up to 0.96 of the cache, 0.82 at
# 1.02), and the chunks are EVEN: a small ragged last chunk runs its steps launch-bound (2.4M members as 1179648 + 1179648 +
# 40704: 0.877; as 2 x 1.2M: 0.909; profiles/r05/chunk_share_sweep.txt, second table).
CHUNK_CACHE_SHARE = 0.7              # (0.8 is as good or better at 4M, 16M and 25M members and 3.5 % worse at 8M: third table of that file)


# The two box-dependent figures the schedules are derived from.  The defaults are what rounds 1-3 measured on MI355X; another
# box (or a future driver) can set them from the environment, or measure them with `calibrate()` below, which overwrites these
# module attributes — engines created afterwards use the new values.
HBM_STREAM_BYTES_PER_S = _env_positive("FIVEEQ_HBM_STREAM_BYTES_PER_S", 6.7e12)    # ceiling of the per-step kernel
LAUNCH_BOUNDARY_S = _env_positive("FIVEEQ_LAUNCH_BOUNDARY_S", 2.0e-6)              # dependent-launch boundary on one stream (measured 1.5-2.6 us)
PER_STEP_SPLIT_MIN_S = 16.0e-6       # a per-step launch is split over two streams from this much traffic time on
PER_STEP_BLOCK = 25                  # steps enqueued per part before switching to the next part's stream
FUSED_SPAN_STEPS = 128               # mode='fused': steps per launch for ensembles of few rounds of waves (see fused_span)
FUSED_SPAN_MAX_ROUNDS = 8.0          # ... up to this many rounds of 4 waves per SIMD
FUSED_SPAN_MIN_ROUNDS = 0.25         # ... and from this many on
# mode='auto' on a LAUNCH-BOUND ensemble (auto_k_steps() > 1) takes the small-ensemble kernel (include/fiveeq.h,
# fiveeq_run_small_*): for a lone 4-pool gas one member per QUAD of lanes while the quads' waves get a SIMD each (one
# 256-thread workgroup per CU: 64 members per CU, 16384 on an MI355X; past that two waves share a SIMD and the unspread form
# is ahead), else one member per lane (fp64 CO2-only, us per step: 0.41 / 0.54 / 0.73 quad / one lane / fused kernel at 10k
# members, 0.89 / 1.09 one lane / fused at 110k, 1.74 / 1.90 at 250k; three gases: 0.94 / 1.38 at 10k, 1.66 / 1.98 at 110k,
# 2.40 / 2.65 at 150k: profiles/r05/small_ensemble_ab.txt, small_ensemble_multigas_ab.txt, auto_window_sweep.txt)
SMALL_QUAD_MEMBERS_PER_CU = 64
# ... and the 4 + 1 + 1 layout one member per OCTET of lanes (round 6, small_octet_kernel: 8 members per wave) up to this many
# members per CU; past it the one-lane form (profiles/r06/small_octet_ab.txt)
SMALL_OCTET_MEMBERS_PER_CU = 64
KSTEPS_LAUNCH_BOUND = 128            # steps per launch of the K-step form on a launch-bound ensemble (2 / 8 / 32 / 128 steps per
                                     # launch at 110k three-gas members: 4.42 / 2.70 / 2.13 / 1.97 us per step; one launch: 1.98)


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


class EnsembleEngine(CheckpointMixin):
    """Advance N ensemble members of the five-equation model on one MI355X."""

    MODES = ("per_step", "graph", "fused", "ksteps", "small", "auto")

    def __init__(self, params, n_members, emissions, *, F_ext=None, dt=1.0, dtype=torch.float64,
                 device=None, store_trajectory=True, output_steps=None, store_concentrations=True,
                 collect_stats=False, hist=None, hist_ring_steps="auto", concentration_driven=False,
                 chunk_members="auto", per_step_streams="auto", fused_span="auto", small_lanes="auto", compensated=False,
                 R0=None, S0=None, lib_path=None):
        """store_trajectory / output_steps: True stores C, T of every step; a list of step indices
        stores only those (rows in increasing step order, see `out_steps`); False stores nothing.
        store_concentrations=False keeps only the T rows (a 100M-member fp32 run then stores 4 B instead
        of 16 B per member and stored step).
        collect_stats: also accumulate per-step ensemble moments of T on the device (`stats()`).
        hist=(lo, hi, n_bins): allocate `T_hist` [n_steps, n_bins] (int64), the fixed-bin histogram of T of
        EVERY step, for all-timestep percentiles (distributed.histogram_percentiles) without a stored
        trajectory.  The stepping kernel also writes each member's BIN INDEX (uint16: 2 bytes per member-step) into a ring of
        `hist_ring_steps` steps, which a histogram pass counts into T_hist — same counts bit for bit as a histogram of
        stored rows:
          run(mode="fused")    two ring slots, the pass drains one on a second HIP stream while the fused kernel fills the
                               other ("auto" ring length: what 8 GB hold, at most 128 steps: 6.4 GB at 12.5M members);
          run(mode="per_step") one slot, counted right behind its steps.
        concentration_driven: inverse mode — `emissions` holds the TARGET concentrations [n_steps, G]
        at the end of each step (shared by all members); the per-member emissions that reach them
        are diagnosed into `self.E` ([n_rows, G, N], aliasing `self.C`), and `self.cumE` [G, N] is
        extra per-member state.  Runs through `run()` as one time-fused launch per call.
        chunk_members: per-step / graph runs of ensembles whose state + parameters exceed the 256 MiB
        Infinity Cache are scheduled chunk-major — all requested steps for members [0, c), then
        [c, 2c), ... — so each chunk's rows stay cache-resident between its consecutive launches
        (0.89-0.91 of 8 TB/s from 4M to 100M members against 0.70-0.79 streamed from HBM; bit-identical results: members
        never interact).  "auto" sizes a chunk's rows to at most CHUNK_CACHE_SHARE of the cache, in even chunks; an int forces c; None / 0 disables it
        (the library then runs such launches with non-temporal row accesses, include/fiveeq.h "CACHE POLICY").
        per_step_streams: mode='per_step' launches each timestep as this many kernels over contiguous member parts, each
        part's launches on its own HIP stream, so that one part's launch tail and ramp overlap the other part's kernel
        (members never interact, so nothing orders the parts against each other).  Two parts: -6.5 % per step at 1M fp64
        members (36.9 -> 34.5 us), -8 % at 0.5M, -2 % at 4M, +10 % at 0.25M (profiles/r03/two_stream_*.txt), bit-identical
        results.  "auto": 2 when one step moves at least ~16 us of traffic, else 1; an int forces it.
        fused_span: mode='fused' covers the requested steps with launches of this many steps (None: one launch).  A SIMD serves
        its oldest wave first, so the waves of a long launch finish in tiers and a launch with FEW rounds of waves ends in a
        long tail; relaunching the same kernel resets the ages (the state crosses HBM once per span: nothing at 128 steps).
        "auto": FUSED_SPAN_STEPS when the ensemble is between FUSED_SPAN_MIN_ROUNDS and FUSED_SPAN_MAX_ROUNDS rounds of resident
        waves, else one launch (profiles/r03/relaunch_sweep.txt).  Bit-identical either way.
        compensated (fp32 only, opt-in): modes 'fused' / 'ksteps' run the COMPENSATED form of the time-fused kernel
        (include/fiveeq.h, fiveeq_run_fused_comp_f32): every pool carries the rounding error of its own update in a second
        register word (no HBM bytes) and the forcing is computed from the excess C - C0 — worst error against 50-digit
        arithmetic C 2.9e-6 -> 1.8e-7, T 1.7e-5 -> 7e-7.  Its own arithmetic: not bit-identical to the default forms; the modes
        that keep the state in HBM between launches ('per_step', 'graph') refuse it; mode 'small' runs it one member per lane
        (fiveeq_run_small_comp_f32), which is what 'auto' takes for a launch-bound ensemble.
        small_lanes: mode='small' (no in-loop histograms): lanes per member, 4 (a lone 4-pool gas: one pool per lane of a
        quad), 8 (the 4 + 1 + 1 layout: one pool per lane of an octet; no collect_stats), 1 (any layout), or "auto" = the widest
        form the layout has while the ensemble is small enough for it (SMALL_*_MEMBERS_PER_CU), else 1."""
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
        self._w = 8 if dtype == torch.float64 else 4
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
        self.small_widest = int(self.lib.fiveeq_small_lanes(G, n_pools))     # 4, 1, or 0 = the layout has no small-ensemble form

        if not store_trajectory:
            output_steps = []
        self.concentration_driven = bool(concentration_driven)
        self.compensated = bool(compensated)
        if self.compensated and (dtype != torch.float32 or self.concentration_driven):
            raise ValueError("compensated=True is an fp32 form of the emission-driven step (fp64 does not need it)")
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
            # per-step (count, sum, sum^2, min, max) folded from the wave records: what a checkpoint's "summaries" carry
            self._step_sums = (torch.zeros((self.n_steps, 5), dtype=torch.float64, device=dev) if collect_stats else None)
            self._step_sums_valid = np.zeros(self.n_steps, dtype=bool)     # steps whose folded sums (above) are current
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
            if hist_ring_steps == "auto":      # as long as 8 GB of 2-byte bin indices allow, at most 128 steps: every chunk
                hist_ring_steps = min(128, max(8, (8 << 30) // (4 * N)))     # boundary is a state + parameter round trip
            self.hist_ring_steps = max(1, min(int(hist_ring_steps), self.n_steps))
            # where the streamed pipeline's histogram pass runs: "side" = a second HIP stream beside the next chunk's fused
            # kernel, "same" = behind each chunk on the caller's stream
            self.hist_pass_stream = _env_choice("FIVEEQ_HIST_PASS_STREAM", "side", ("side", "same"))
            self._bins = None            # the bin-index ring, allocated by the first run that fills T_hist
        if chunk_members == "auto":
            chunk_members = self.auto_chunk(N, SP, G, dtype)
        self.chunk_members = int(chunk_members or 0) // 256 * 256
        if per_step_streams == "auto":
            t_step = min(N, self.chunk_members or N) * self._w * (2 * SP + 4 * G + 7) / HBM_STREAM_BYTES_PER_S
            per_step_streams = 2 if t_step >= PER_STEP_SPLIT_MIN_S else 1
        self.per_step_streams = max(1, int(per_step_streams))
        if fused_span not in ("auto", None) and int(fused_span) < 1:
            raise ValueError("fused_span must be 'auto', None or a positive number of steps")
        self.fused_span = fused_span if fused_span in ("auto", None) else int(fused_span)
        if small_lanes != "auto" and int(small_lanes) not in (1, 4, 8):
            raise ValueError("small_lanes must be 'auto', 1, 4 or 8")
        self.small_lanes = small_lanes if small_lanes == "auto" else int(small_lanes)
        # run(..., join=False) left work nobody has waited for: the streams of THAT run ([its main, its side streams]) — join()
        # waits for exactly these, whichever stream the consumer is on; None = nothing outstanding
        self._ps_unjoined = None
        self._R0 = None if R0 is None else np.asarray(R0, dtype=np.float64).reshape(SP, N)
        self._S0 = None if S0 is None else np.asarray(S0, dtype=np.float64).reshape(2, N)
        self.t_next = 0                     # first step not yet run (bookkeeping for checkpoints)
        self.last_mode = None               # the mode the last run() used after resolving 'auto'
        self.reset_state()
        self._plans = {}
        with torch.cuda.device(self.device):
            self.probe_streams()            # construction is synchronous anyway: the one place the probe may clock streams

    @staticmethod
    def auto_chunk(n_members, sum_pools, n_gas, dtype, cache_bytes=INFINITY_CACHE_BYTES):
        """Members per chunk of the chunk-major schedule: the fewest EVEN chunks whose state + parameter rows take at most
        CHUNK_CACHE_SHARE of the Infinity Cache each (a multiple of 256 members); 0 = do not chunk: the ensemble's rows fit the
        cache by themselves."""
        w = 8 if dtype == torch.float64 else 4
        rows_bytes = n_members * w * (sum_pools + 2 + 3 * n_gas + 2)
        if rows_bytes <= cache_bytes:
            return 0
        k = -(-rows_bytes // int(CHUNK_CACHE_SHARE * cache_bytes))
        return -(-n_members // (256 * k)) * 256

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
        """Steps per launch of the per-step family: 1 (the per-step kernel, the north-star form) while one step's HBM traffic
        hides the dependent-launch boundary (at least three boundaries' worth: ~160k three-gas fp64 members), otherwise —
        the ensemble is launch-bound, not bandwidth-bound — KSTEPS_LAUNCH_BOUND: state and parameters cross HBM once per
        launch, so nothing is gained by a shorter span (round 4 scaled K with the ensemble and ran 110k members at K = 2)."""
        t_step = self.n_members * self.bytes_per_member_step("per_step") / HBM_STREAM_BYTES_PER_S
        return 1 if t_step >= 3.0 * LAUNCH_BOUNDARY_S else min(KSTEPS_LAUNCH_BOUND, self.n_steps)

    def small_form(self):
        """Lanes per member mode='small' would run with now (4 or 1); 0 = the small-ensemble kernel does not apply: a run that
        wants in-loop histograms or the concentration-driven form."""
        if not self.small_widest or self.T_hist is not None or self.concentration_driven:
            return 0
        if self.compensated:                                     # fiveeq_run_small_comp_f32: one member per lane, every layout
            return 1 if self.small_lanes in ("auto", 1) else 0
        octet_ok = self.small_widest == 8 and not self.collect_stats       # the octet form writes no per-wave statistics
        if self.small_lanes != "auto":
            if self.small_lanes == 8:
                return 8 if octet_ok else 0
            return self.small_lanes if self.small_lanes in (1, self.small_widest) else 0
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        if self.small_widest == 4 and self.n_members <= SMALL_QUAD_MEMBERS_PER_CU * cus:
            return 4
        if octet_ok and self.n_members <= SMALL_OCTET_MEMBERS_PER_CU * cus:
            return 8
        return 1

    def resolve_mode(self, mode, k_steps=None):
        """(mode, k_steps) run() uses for a request: 'auto' resolved, everything else as given.  'auto' is 'per_step' (the
        north-star form) while a step's HBM traffic hides the launch boundary; a launch-bound ensemble takes the small-ensemble
        kernel (small_form()) — or, with hist=, 'fused' (the streamed histogram pipeline, the fastest form that fills T_hist
        there: profiles/r04/auto_hist_table.txt); an explicit k_steps, or a run the small kernel does not serve, 'ksteps'."""
        if mode != "auto":
            return mode, k_steps
        if self.compensated:                                     # the compensation words live in registers: the time-fused kernel,
            launch_bound = (self.auto_k_steps() if k_steps is None else int(k_steps)) > 1     # or the small-ensemble one (one lane)
            return ("small" if launch_bound and k_steps is None and self.small_form() else "fused"), None
        k = self.auto_k_steps() if k_steps is None else int(k_steps)
        if k <= 1:
            return "per_step", None
        if self.T_hist is not None:
            return "fused", None
        if k_steps is None and self.small_form():
            return "small", None
        return "ksteps", k

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

    def _wave_stats(self):
        """The per-wave record buffer of the in-kernel statistics, allocated by the first launch that writes it."""
        if self.collect_stats and self.T_stats is None:
            self.T_stats = torch.zeros((self.n_waves, self.n_steps, 4), dtype=torch.float64, device=self.device)
        return self.T_stats

    # -- launches ----------------------------------------------------------------------
    def _stream(self, stream=None):
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        return ctypes.c_void_p(s.cuda_stream)

    def _ptr(self, t, byte_off=0):
        return ctypes.c_void_p(0 if t is None else t.data_ptr() + byte_off)

    def _run_args(self, t_begin, t_end, m0=0, n=None):
        """C-ABI arguments (model ... T_stats) for members [m0, m0 + n) of this engine's rows (ld = N)."""
        N, w = self.n_members, self._w
        n = N if n is None else n
        return (ctypes.byref(self.model), n, N, self._ptr(self.drive), self.n_steps, int(t_begin), int(t_end),
                self._ptr(self.r, m0 * w), self._ptr(self.q, m0 * w), self._ptr(self.R, m0 * w), self._ptr(self.S, m0 * w),
                self._ptr(self.C, m0 * w), self._ptr(self.T, m0 * w), self.n_rows,
                self._ptr(self.T_stats, (m0 // 64) * self.n_steps * 4 * 8))

    def _fn(self, name):
        return getattr(self.lib, f"fiveeq_{name}_{self._sfx}")

    def _chunks(self):
        N, c = self.n_members, self.chunk_members
        if not c or c >= N:
            return [(0, N)]
        return [(m0, min(c, N - m0)) for m0 in range(0, N, c)]

    def _run_inverse(self, t_begin, t_end, stream):
        a = self._run_args(t_begin, t_end)
        return self._fn("run_inverse")(*a[:11], self._ptr(self.cumE), *a[11:], self._stream(stream))

    def step(self, t, stream=None):
        """One timestep = one kernel launch (asynchronous)."""
        t = int(t)
        if self._ps_unjoined:                         # a run(..., join=False) may still be writing R, S, T_stats on the side streams
            self.join(stream)
        self._step_sums_valid[t] = False              # this launch writes the step's wave record: older folded moments are stale
        if self.collect_stats:
            self._stats_have[t] = True
        self._wave_stats()
        with torch.cuda.device(self.device):
            if self.concentration_driven:
                rc = self._run_inverse(t, t + 1, stream)
            else:
                a = self._run_args(t, t + 1)
                rc = self._fn("step")(*a[:5], t, *a[7:], self._stream(stream))
        _capi.check(self.lib, rc)
        self.t_next = t + 1

    def run(self, t_begin=0, t_end=None, mode="per_step", stream=None, k_steps=None, join=True):
        """Advance steps [t_begin, t_end).  mode:
        'per_step' one launch per timestep, enqueued from C (the north-star form);
        'graph'    the same launches replayed from a captured hipGraph;
        'fused'    one launch, state in registers across all steps (with `hist=`: chunks of hist_ring_steps steps, T_hist
                   filled by the streamed pipeline, see __init__);
        'ksteps'   the fused kernel over consecutive spans of `k_steps` steps (default `auto_k_steps()`):
                   state crosses HBM once per k_steps — the per-step family's answer for launch-bound ensembles;
        'small'    the small-ensemble kernel (see small_lanes): the model in registers, one launch; a lone 4-pool gas: one member
                   per quad of lanes;
        'auto'     see resolve_mode().
        Every mode gives bit-identical results.
        join=False (mode 'per_step' on several streams only): do not make the caller's stream wait for the side streams at
        the end, and do not make the side streams wait for the caller's stream at the start of the NEXT such call — for
        back-to-back calls with nothing in between that touches the state on the caller's stream (bench.py's repeated
        blocks): a join is two cross-stream hops, ~20 us.  Call `join()` before anything consumes the results."""
        t_begin, t_end = int(t_begin), self.n_steps if t_end is None else int(t_end)
        if self._ps_unjoined and mode != "per_step":
            self.join(stream)
        mode, k_steps = self.resolve_mode(mode, k_steps)
        self.last_mode = mode            # what 'auto' resolved to (tests, bench.py's config.mode)
        if mode not in self.MODES:
            raise ValueError(f"unknown mode {mode!r}")
        if self.T_hist is not None and mode not in ("fused", "per_step"):
            raise ValueError(f"mode {mode!r} does not fill T_hist: use 'fused' or 'per_step' with hist=")
        if self.compensated and mode not in ("fused", "ksteps", "small"):
            raise ValueError(f"mode {mode!r} has no compensated form: the compensation words live in registers, so only the "
                             "time-fused kernel ('fused', 'ksteps') and the small-ensemble kernel ('small', one lane) carry them")
        if mode == "small" and not self.small_form():
            raise ValueError("mode 'small' serves runs without in-loop histograms or the inverse form, with 4 lanes per "
                             "member for a lone 4-pool gas only and 8 for pools [4, 1, 1] without collect_stats "
                             f"(pools {self.pools}, small_lanes={self.small_lanes!r})")
        with torch.cuda.device(self.device):
            self._wave_stats()
            self._step_sums_valid[t_begin:t_end] = False     # these launches write per-wave records: older folded sums are stale
            if self.collect_stats:
                self._stats_have[t_begin:t_end] = True
            if self.concentration_driven:
                rc = self._run_inverse(t_begin, t_end, stream)
            elif mode == "per_step":
                rc = self._run_per_step(t_begin, t_end, stream, join)
            elif mode == "fused" and self.T_hist is not None:
                rc = self._run_fused_bin_ring(t_begin, t_end, stream)
            elif self.compensated and mode == "small":
                rc = self.lib.fiveeq_run_small_comp_f32(*self._run_args(t_begin, t_end), self._stream(stream))
            elif self.compensated:                                   # 'fused' / 'ksteps' without a ring: one C call
                span = (self.fused_span_steps(t_end - t_begin) if mode == "fused" else
                        max(self.auto_k_steps() if k_steps is None else int(k_steps), 1))
                rc = self.lib.fiveeq_run_fused_comp_f32(*self._run_args(t_begin, t_end), span, 0.0, 1.0, 1, None, 0,
                                                        self._stream(stream))
            elif mode == "fused":
                span = self.fused_span_steps(t_end - t_begin)
                if span < t_end - t_begin:                       # the same kernel, relaunched every `span` steps
                    rc = self._fn("run_ksteps")(*self._run_args(t_begin, t_end), span, self._stream(stream))
                else:
                    rc = self._fn("run_fused")(*self._run_args(t_begin, t_end), self._stream(stream))
            elif mode == "ksteps":
                k = self.auto_k_steps() if k_steps is None else int(k_steps)
                rc = self._fn("run_ksteps")(*self._run_args(t_begin, t_end), max(k, 1), self._stream(stream))
            elif mode == "small":
                rc = self._fn("run_small")(*self._run_args(t_begin, t_end), self.small_form(), self._stream(stream))
            else:                                                # 'graph': one captured plan per (chunk, part), the parts of a
                plans = self.prepare_graph(t_begin, t_end)       # chunk replayed side by side on their own streams
                rc = self._on_part_streams(stream, True, lambda streams: self._first_error(
                    self.lib.fiveeq_plan_launch(plan, self._stream(streams[si]))
                    for plan, (_, _, si) in zip(plans, self.per_step_launches())))
        _capi.check(self.lib, rc)
        self.t_next = t_end

    @staticmethod
    def _first_error(codes):
        """The first non-zero return code of a sequence of C calls, which stops at it (a generator: later calls are not made)."""
        for rc in codes:
            if rc != _capi.OK:
                return rc
        return _capi.OK

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
        if n_streams == 1:
            return [main]
        # side streams PROBED to run beside `main` (a stream that shares main's hardware queue serialises the parts, +12 % per
        # step): probed at construction / by probe_streams(), never here — an unprobed main gets plain streams (tuning.py)
        return [main] + concurrent_side_streams(self.lib, main, n_streams - 1)

    def join(self, stream=None):
        """Make `stream` (default: the current stream) wait for everything run(..., join=False) left outstanding — on the
        streams THAT run used: its side streams and, when the consumer is another stream than the run's main, that main too."""
        pending, self._ps_unjoined = self._ps_unjoined, None
        if pending:
            consumer = stream if stream is not None else torch.cuda.current_stream(self.device)
            for s in pending:
                if s.cuda_stream != consumer.cuda_stream:
                    consumer.wait_stream(s)

    def probe_streams(self, stream=None):
        """Probe (tuning.concurrent_side_streams, probe=True) side streams for `stream` (default: the current one) so that the
        two-stream schedules launched from it really overlap.  SYNCHRONISES that stream: call it where that is harmless — the
        constructor does, for the stream current at construction; a caller that runs the engine on another stream of its own
        calls this once before its first run().  Returns side_stream_report()."""
        main = stream if stream is not None else torch.cuda.current_stream(self.device)
        want = self._side_streams_wanted()
        if want:
            if self._ps_unjoined:
                self.join(main)
            concurrent_side_streams(self.lib, main, want, probe=True)
        return self.side_stream_report(main)

    def _side_streams_wanted(self):
        n = max(i for _, _, i in self.per_step_launches())
        if self.T_hist is not None and self.hist_pass_stream == "side":
            n = max(n, 1)
        return n

    def side_stream_report(self, stream=None):
        """{"wanted", "probed", "passed", "candidates_tried", "probe_enabled"} for the side streams runs on `stream` use."""
        main = stream if stream is not None else torch.cuda.current_stream(self.device)
        want = self._side_streams_wanted()
        return {"wanted": want, **side_stream_report(main, want)}

    def _on_part_streams(self, stream, join, body):
        """Run body(streams) with the side streams ordered behind the caller's stream before it and (join=True) the caller's
        stream ordered behind them after it — the frame of every form that runs the parts of a chunk side by side."""
        streams = self.per_step_stream_list(stream)
        if self._ps_unjoined and [s.cuda_stream for s in self._ps_unjoined] != [s.cuda_stream for s in streams]:
            self.join(streams[0])                                # an unjoined run on OTHER streams: order this one behind it
        if not self._ps_unjoined:
            for s in streams[1:]:
                s.wait_stream(streams[0])                        # the state may have been touched on the caller's stream
        rc = body(streams)
        if len(streams) > 1:
            if join:
                for s in streams[1:]:
                    streams[0].wait_stream(s)
            self._ps_unjoined = None if join else list(streams)
        return rc

    def _per_step_schedule(self, t_begin, t_end, stream, join, block, launch):
        """The launch order of the per-step family: member chunks one after the other (chunk-major), within a chunk blocks of
        steps [t, t1) — t1 the next multiple of `block` past t (block = None: the whole range at once) — and within a block
        the parts of the chunk side by side, `launch(t, t1, m0, n, stream)` for each on its own stream."""
        layout = self.per_step_launches()
        firsts = [i for i, (_, _, si) in enumerate(layout) if si == 0] + [len(layout)]

        def body(streams):
            for a, b in zip(firsts[:-1], firsts[1:]):
                t = t_begin
                while t < t_end:
                    t1 = t_end if block is None else min(t_end, (t // block + 1) * block)
                    for m0, n, si in layout[a:b]:
                        rc = launch(t, t1, m0, n, streams[si])
                        if rc != _capi.OK:
                            return rc
                    t = t1
            return _capi.OK

        return self._on_part_streams(stream, join, body)

    def _run_per_step(self, t_begin, t_end, stream, join=True):
        """fiveeq_run_*: one launch per timestep and member part, enqueued from C.  With T_hist: fiveeq_run_bins_*, whose
        kernel also writes every member's histogram bin into a ring strip [S, N] of uint16 (row t mod S); after S steps each
        part counts its strip into T_hist (fiveeq_hist_bins) on its own stream — 2 bytes written + 2 read per member-step on
        top of the step's 124 / 248."""
        if self.T_hist is None:
            # several streams: short blocks keep every stream's queue fed; one stream: one C call per chunk
            block = PER_STEP_BLOCK if self.per_step_streams > 1 else None
            run = self._fn("run")
            return self._per_step_schedule(t_begin, t_end, stream, join, block,
                                           lambda t, t1, m0, n, s: run(*self._run_args(t, t1, m0, n), self._stream(s)))
        N, (lo_h, hi_h, nb) = self.n_members, self.hist_spec
        ring = self._bin_ring(slots=1)
        S, buf, run = ring["S"], ring["buf"][0], self._fn("run_bins")

        def launch(t, t1, m0, n, s):
            st = self._stream(s)
            return (run(*self._run_args(t, t1, m0, n), lo_h, hi_h, nb, self._ptr(buf, m0 * 2), S, st)
                    or self.lib.fiveeq_hist_bins(t1 - t, n, N, self._ptr(buf[t % S], m0 * 2), nb, self._ptr(self.T_hist[t:t1]), st))

        return self._per_step_schedule(t_begin, t_end, stream, True, S, launch)

    def _bin_ring(self, slots=2):
        """Ring [slots, S, N] of uint16 bin indices for the streamed histograms: mode 'fused' double-buffers (two slots), mode
        'per_step' counts each strip right behind its steps (one slot: half the memory); grown to two slots when a fused
        run follows a per-step one, rebuilt when hist_ring_steps changed."""
        N, S, dev = self.n_members, max(1, min(int(self.hist_ring_steps), self.n_steps)), self.device
        ring = self._bins
        if ring is None or ring["S"] != S or ring["buf"].shape[0] < slots:
            torch.cuda.synchronize(dev)
            self._bins = None
            self._bins = ring = {"S": S, "buf": torch.empty((slots, S, N), dtype=torch.int16, device=dev),
                                 "drained": [torch.cuda.Event(), torch.cuda.Event()]}
        return ring

    def _run_fused_bin_ring(self, t_begin, t_end, stream):
        """mode='fused' with hist=: chunks of S = hist_ring_steps steps.  Stream A (the caller's) runs
        fiveeq_run_fused_bins_* for chunk i: the ordinary fused kernel — stored C/T rows and per-wave statistics exactly as
        in a plain fused run — that also writes every member's histogram bin of every step into ring slot i % 2 (row t mod
        S).  Stream B waits for the chunk and counts the slot's rows into T_hist[t:t+S] (fiveeq_hist_bins); A reuses a slot
        only after B has drained it."""
        N, dev = self.n_members, self.device
        ring = self._bin_ring()
        S = ring["S"]
        main = stream if stream is not None else torch.cuda.current_stream(dev)
        side = main if self.hist_pass_stream == "same" else concurrent_side_streams(self.lib, main, 1)[0]
        side.wait_stream(main)
        lo_h, hi_h, nb = self.hist_spec
        if self.compensated:                                     # one launch per chunk either way: k_steps = the chunk
            fused = lambda *a: self.lib.fiveeq_run_fused_comp_f32(*a[:15], a[6] - a[5], *a[15:])     # noqa: E731
        else:
            fused = self._fn("run_fused_bins")
        used = [False, False]
        rc, t, i = _capi.OK, t_begin, 0
        while t < t_end and rc == _capi.OK:
            t1 = min(t_end, (t // S + 1) * S)              # chunks end on multiples of S: row = t mod S never wraps
            slot = i % 2
            buf = ring["buf"][slot]
            if used[slot]:
                main.wait_event(ring["drained"][slot])
            rc = fused(*self._run_args(t, t1), lo_h, hi_h, nb, self._ptr(buf), S, self._stream(main))
            side.wait_stream(main)
            if rc == _capi.OK:
                rc = self.lib.fiveeq_hist_bins(t1 - t, N, N, self._ptr(buf[t % S:]), nb, self._ptr(self.T_hist[t:t1]),
                                               self._stream(side))
                ring["drained"][slot].record(side)
                used[slot] = True
            t, i = t1, i + 1
        main.wait_stream(side)
        return rc

    def prepare_graph(self, t_begin=0, t_end=None):
        """Capture (once) the per-step launches of [t_begin, t_end) into hipGraph plans, one per launch of
        `per_step_launches()` (member chunk x part); returns the list of plans in that order."""
        t_end = self.n_steps if t_end is None else int(t_end)
        key = (int(t_begin), t_end)
        plans = self._plans.get(key)
        if plans is None:
            plans = []
            self._wave_stats()
            fn = self._fn("plan_create")
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
        per-wave records the kernels wrote (or, for steps a checkpoint's summaries brought, as folded by the saver).
        Additive across shards (fiveeqscm_amd.distributed.reduce_stats)."""
        if not self.collect_stats:
            raise RuntimeError("engine was built with collect_stats=False")
        if self._ps_unjoined:
            self.join()
        t_end = self.n_steps if t_end is None else int(t_end)
        valid = self._step_sums_valid[t_begin:t_end]
        if valid.all():                                          # everything came folded from a checkpoint
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

    def gather_summary(self, steps, percentiles=(5.0, 50.0, 95.0), dst=0, group=None, stats=None, gas=None):
        """End-of-run summary of T — or, with `gas` = a gas index, of that gas's concentration C — at the stored `steps` over
        ALL members of all ranks (collective over `group`; see distributed.gather_summary): merged moments on every rank,
        exact percentiles on rank `dst`.  With collect_stats the moments of T come from the records the kernels wrote while
        stepping — the summary then reads the rows twice (histogram, selection) instead of three times."""
        from .distributed import gather_summary
        stored = self.T if gas is None else self.C
        if stored is None or self.concentration_driven and gas is not None:
            raise RuntimeError(f"no stored {'T' if gas is None else 'C'} rows to summarise")
        if gas is not None and not 0 <= int(gas) < self.n_gas:
            raise ValueError(f"gas {gas}: the parameter set has gases 0..{self.n_gas - 1}")
        if self._ps_unjoined:
            self.join()
        row_of = {int(t): r for r, t in enumerate(self.out_steps)}
        missing = [int(t) for t in steps if int(t) not in row_of]
        if missing:
            raise ValueError(f"steps {missing} are not stored (out_steps)")
        picked = [row_of[int(t)] for t in steps]
        rows = self.T[picked] if gas is None else self.C[picked, int(gas)]
        sums = None
        if gas is None and self.collect_stats and all(self._stats_have[int(t)] for t in steps):   # else: the moments pass over the rows
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
        with torch.cuda.device(self.device):
            rc = self._fn("hist_rows")(k, self.n_members, x.shape[1], self._ptr(x), float(lo), float(hi), int(n_bins),
                                       self._ptr(out), self._stream(stream))
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
        fused / small:   w (G + 1) + w (2 SP + 3 G + 6) / steps per launch (n_steps; fused: fused_span, or with hist= the
                         ring length);
        ksteps:          w (G + 1) + w (2 SP + 3 G + 6) / k_steps  (state + parameters once per k_steps).
        Statistics add one 32-byte record per wave of 64 members and step; with `hist=` the bin ring adds 2 B written + 2 B
        read per member-step."""
        w, G, SP = self._w, self.n_gas, self.sum_pools
        out = ((G if self.C is not None else 0) + 1) * self.n_rows / self.n_steps      # stored rows only
        extra = (32.0 / 64.0) if self.collect_stats else 0.0
        ring = 4.0 if (self.T_hist is not None and mode in ("fused", "per_step")) else 0.0
        if mode == "per_step":
            return w * (2 * SP + 3 * G + 6 + out) + extra + ring
        if mode == "fused":
            span = min(self.hist_ring_steps, self.n_steps) if self.T_hist is not None else self.fused_span_steps(self.n_steps)
        elif mode == "ksteps":
            span = k_steps or self.auto_k_steps()
        elif mode == "small":
            span = self.n_steps
        else:
            raise ValueError(f"no byte count for mode {mode!r}")
        return w * (out + (2 * SP + 3 * G + 6) / max(int(span), 1)) + extra + ring


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
