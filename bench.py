#!/usr/bin/env python3
"""bench.py — ensemble-member-timesteps/sec of the five-equation engine on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Both forms work for every N: called plainly with N > 1, this script starts the launcher form itself as a CHILD process
before it imports torch or touches a GPU (`_self_launch`), relays the child's one line and exits with its code.

A "step" is ONE MODEL TIMESTEP of the whole ensemble = one launch of the per-timestep HIP kernel
(fiveeq_step_f64, include/fiveeq.h) on each GPU.  Workload (BASELINE.json configs[2], the
largest single-GPU configuration and the one the roofline target is quoted for): 1,000,000
members per GPU (weak scaling), CO2+CH4+N2O, fp64, deterministic RCP-like emissions
(SURVEY.md section 8d), Latin-hypercube parameter draws (seed 20261003; the shard-computable design:
every rank draws exactly its own members on its own GPU), state and parameters resident in HBM, C/T
trajectory rows written every step.  Timesteps cycle through the 750-step scenario (t = k mod 750); the
default K + W = 750 is exactly one scenario pass.

Rank 0 prints ONE JSON line.  `value` = (members on all GPUs) x K / the K-step block time, MAX over ranks: the block is
clocked once on the wall clock (barrier, device sync, clock, K steps, drain, clock: `timing.first_block_ms_per_step`) and then
repeated back to back for --timed-s (6.5) seconds of device time, HIP event to HIP event; the MEDIAN block is reported.
`timing.host_enqueue_us_per_step` / `host_share` say how long the rank's host thread needs to enqueue a step with all ranks
enqueuing at once.  `roofline.traffic` (N = 1, per-step form) is MEASURED BY THE RUN ITSELF: two child `rocprofv3 --pmc` passes
(FETCH_SIZE, WRITE_SIZE) of the same kernel at the same size before this process touches the GPU, calibrated on a known copy.  `roofline` prices the kernel of the chosen --mode: the per-step kernel against the
8 TB/s HBM peak with the ALGORITHMIC bytes A = w(2 SP + 4 G + 7) = 248 B per member-step (plus the same
kernel on an ensemble far beyond the Infinity Cache, `hbm_resident`, and its fp64 VALU issue fraction);
the fused / K-step / small-ensemble kernels with their own A and bound "fp64-valu"; `roofline.single_launch_*` is the per-step
kernel as ONE launch per timestep on one stream (north_star's literal shape) beside the default two-launch form.  `cpu_baseline` times the CPU oracle
(NumPy, one process per usable core, and the plain-C port under OpenMP) on this box's host cores on a bounded sample
(rank 0, N=1 only) BEFORE the GPU is touched, so that the GPU work of the run is one contiguous window.

The default launch form is the per-step one at EVERY N, so that the driver's N = 1, 2, 4, 8 values compare like with like.  With
N > 1 the line also says whether the slowest rank's host thread needs more than --host-share-limit (0.5) of a step to enqueue it
(`timing.host_fallback.would_switch`); only with --host-fallback does the timed region then switch to the hipGraph replay of the
same launches (same kernels, same bits): `config.mode` says what ran.

N > 1: one process per GPU; time-stepping needs no collective.  The line proves what ran: `config.devices` (every rank's device
index, name, PCI bus id, uuid), `timing.per_rank_ms_per_step`, `timing.per_rank_host_enqueue_us` — gathered over the control
plane, so they survive an RCCL failure — and, inside the protected summary section, `summary.rccl_world_size` /
`summary.backend_seen` as the data group itself reports them after its first collective, and the bytes every rank sent the root.  The barriers around the clocked region and the MAX of the
clocked times go over a gloo control group (host scalars); the end-of-run summary exchange — the only collective that moves
ensemble data — goes over RCCL, LAST, with the line already complete and a watchdog thread beside it: an RCCL exchange that FAILS
(first contact across xGMI refusing to come up) is repeated over the gloo control plane — the line then carries the summary AND
`summary.rccl_error`, exit 0; one that HANGS costs the line its `summary` (-> {"error": ...}, non-zero exit), never its measurement.  Every rank's host thread is bound
to the CPUs next to its GPU (fiveeqscm_amd/hostbind.py: sysfs, os.sched_setaffinity, BEFORE the first GPU call; no wrapper, no
re-exec) and says so in `config.devices[].cpus` / `.cpu_binding`; `.side_streams` says whether the side stream of the two-launch
per-step form passed its concurrency probe.

Layout of this file: arguments; main() = set-up, THE CLOCKED REGION (timed_block / clock_blocks), the line, the end-of-run exchange;
then what runs before the GPU is touched (self-launch, CPU baseline).  The measurement legs the line reports beside `value`
(roofline batches, copy rates, the beyond-the-cache leg, host enqueue time) live in benchlib/legs.py, the gloo control plane in
benchlib/control.py, the child rocprofv3 passes behind `roofline.traffic` in benchlib/traffic.py.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

N_SCEN = 750            # steps of the scenario; timesteps cycle through it
WORKLOADS = {
    # name: (param set, gases, members per GPU, description)
    "config2": ("co2", 1, 10_000, "BASELINE configs[1]: 10k-member CO2-only ensemble, perturbed r0/rC/rT + TCR/ECS"),
    "config3": ("multigas", 3, 1_000_000, "BASELINE configs[2]: 1M-member CO2+CH4+N2O ensemble per GPU"),
    "config4": ("multigas", 3, 1_250_000, "BASELINE configs[3]: 10M-member multi-gas ensemble over 8 GPUs (1.25M/GPU)"),
    "config5": ("multigas", 3, 12_500_000, "BASELINE configs[4]: 100M-member multi-gas ensemble over 8 GPUs (12.5M/GPU)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=740)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--members", type=int, default=0, help="members per GPU (default: the workload's)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--mode", default=None, choices=["per_step", "graph", "fused", "ksteps", "small", "auto"],
                    help="default: per_step at every N (see --host-fallback)")
    ap.add_argument("--host-share-limit", type=float, default=0.5)
    ap.add_argument("--host-fallback", action="store_true",
                    help="N > 1 without --mode: switch the timed region to graph replay when the per-step enqueue share reaches "
                         "--host-share-limit (default: measure and report it, keep the per-step form)")
    ap.add_argument("--k-steps", type=int, default=0, help="steps per launch for --mode ksteps (0: the engine's choice)")
    ap.add_argument("--no-trajectory", action="store_true", help="do not store C/T rows (drops G+1 writes from A)")
    ap.add_argument("--compensated", action="store_true",
                    help="--dtype f32 --mode fused|ksteps only: the compensated fp32 form of the time-fused kernel (include/fiveeq.h)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic in this run (two child rocprofv3 --pmc passes before the GPU is touched, "
                         "N = 1 and the per-step form only): take the committed profiles/traffic.json figure, labelled as such")
    ap.add_argument("--no-hbm-resident", action="store_true", help="skip the beyond-Infinity-Cache roofline leg")
    ap.add_argument("--hbm-resident-members", type=int, default=8_000_000)
    ap.add_argument("--hbm-placements", type=int, default=5,
                    help="placements of the stored-trajectory buffers the beyond-the-cache leg measures (median reported)")
    ap.add_argument("--cpu-sample-members", type=int, default=1_500_000)
    ap.add_argument("--kernel-batches", type=int, default=5, help="event-timed batches of 100 launches for roofline")
    ap.add_argument("--timed-s", type=float, default=6.5,
                    help="device seconds to clock: the K-step block is repeated back to back until this much device time has "
                         "been clocked and the MEDIAN block is reported (0: time ONE block on the wall clock)")
    ap.add_argument("--max-repeats", type=int, default=20001)
    ap.add_argument("--dist-timeout-s", type=float, default=120.0, help="timeout of every process group (init and collectives)")
    ap.add_argument("--summary-watchdog-s", type=float, default=150.0,
                    help="N > 1: if the end-of-run exchange has not returned after this long, rank 0 prints the (complete) "
                         "line with summary.error and the job exits non-zero")
    ap.add_argument("--launch-timeout-s", type=float, default=570.0,
                    help="plain `--gpus N` form only: the self-started child job is ended after this long")
    ap.add_argument("--numpy-baseline", action="store_true",
                    help="add SURVEY 8d's NumPy legs to cpu_baseline: N = 1e5 on one core and on one process per usable "
                         "core, CO2-only and multi-gas (baseline only; adds ~1-2 min)")
    return ap.parse_args()


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:       # only reachable when bench.main() is called from other code: the script self-launches
            sys.exit(f"--gpus {a.gpus} needs WORLD_SIZE / RANK / LOCAL_RANK from a launcher (run bench.py as a script)")
        a.gpus = world
    kind, G, per_gpu, desc = WORKLOADS[a.workload]
    per_gpu = a.members or per_gpu
    n_total = per_gpu * world

    # ---- CPU baseline FIRST (rank 0 of a one-GPU run only): the GPU has not been touched yet, so everything after this leg
    # is one contiguous window of GPU work for whoever samples the card from outside ------------------------------------
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(kind, G, a.cpu_sample_members, N_SCEN, numpy_legs=a.numpy_baseline)

    # ---- roofline.traffic MEASURED IN THIS RUN (rank 0 of a one-GPU per-step run): two child `rocprofv3 --pmc` passes, FETCH_SIZE and
    # WRITE_SIZE separately, of the same kernel at the same size (tools/pmc_workload.py), each calibrated on a known copy in the
    # same pass — as child processes BEFORE this process touches the GPU ---------------------------------------------------
    traffic_live = None
    if (rank == 0 and world == 1 and not a.no_live_traffic and not a.no_trajectory and (a.mode or "per_step") in ("per_step", "graph")
            and "rocprof" not in os.environ.get("LD_PRELOAD", "")):      # (not from inside a profiled process)
        from benchlib.traffic import live_traffic          # (no torch in there: the GPU stays untouched by this process)
        traffic_live = live_traffic(kind, a.dtype, per_gpu)

    # ---- this rank's host thread goes next to its GPU — read from sysfs and applied BEFORE anything touches the GPU (no wrapper,
    # no re-exec; the HIP runtime's helper threads inherit the mask); checked against the runtime's own PCI address below -----
    from fiveeqscm_amd import hostbind
    try:
        binding = hostbind.bind_rank(local_rank, local_world)
    except Exception as exc:  # noqa: BLE001 - a binding that cannot be worked out must never cost the measurement
        binding = {"applied": False, "reason": f"{type(exc).__name__}: {exc}"}

    import torch
    from benchlib import legs
    from benchlib.control import Control
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # FIVEEQ_BENCH_BACKEND=gloo rehearses the multi-process path on a box with fewer GPUs than ranks
    # (ranks share devices, the summary exchange goes through host memory); the default is RCCL.
    backend = os.environ.get("FIVEEQ_BENCH_BACKEND", "nccl")
    # one rank per GPU; more ranks than GPUs share the devices round-robin — the gloo rehearsals on a one-GPU box do that on
    # purpose, and under RCCL it is a misconfiguration that RCCL itself reports on first contact ("duplicate GPU"), i.e. inside
    # the watchdog-protected summary section: the measurement survives (tested)
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    # FIVEEQ_BENCH_FORCE_DIST=1: build the process groups and run every collective of the N > 1 path (barriers, the MAX
    # of the block times, the summary exchange) in a ONE-rank job too — RCCL first contact for this file on a one-GPU box.
    ctl = Control(world, rank, dev, a.dist_timeout_s, force=world == 1 and os.environ.get("FIVEEQ_BENCH_FORCE_DIST") == "1")
    dist = ctl.dist

    from fiveeqscm_amd import emissions, params
    from fiveeqscm_amd.distributed import gather_summary, shard_bounds
    from fiveeqscm_amd.engine import EnsembleEngine

    dtype = torch.float64 if a.dtype == "f64" else torch.float32
    k_steps = a.k_steps or None

    # This rank's contiguous shard [lo, hi) of ONE Latin hypercube over all members, drawn on this
    # rank's GPU: O(shard) work and memory whatever the world size, identical design for any world size.
    t_setup = time.perf_counter()
    lo, hi = shard_bounds(n_total, rank, world)
    n_local = hi - lo
    p = params.sample_ensemble_shard(params.default_params(kind), n_total, lo, hi, device=dev, dtype=dtype)
    E = emissions.rcp_like_emissions(N_SCEN, G)
    eng = EnsembleEngine(p, n_local, E, dtype=dtype, device=dev, store_trajectory=not a.no_trajectory, compensated=a.compensated)
    torch.cuda.synchronize(dev)
    setup_s = time.perf_counter() - t_setup

    props = torch.cuda.get_device_properties(dev)
    pci = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0))
    try:
        binding = hostbind.verify(binding, pci, local_rank, local_world)
    except Exception as exc:  # noqa: BLE001
        binding = dict(binding, verify_error=f"{type(exc).__name__}: {exc}")
    devices = ctl.gather_over_ranks({"rank": rank, "local_rank": local_rank, "device_index": dev_index, "name": props.name,
                                     "pci_bus_id": pci, "uuid": str(getattr(props, "uuid", "")), "pid": os.getpid(),
                                     "visible_devices": torch.cuda.device_count(), "cpus": binding.get("cpus"),
                                     "cpu_binding": binding, "side_streams": eng.side_stream_report()})
    legs.spin_up(eng, dev)

    # ---- warm-up, then EXACTLY K timed steps ------------------------------------------------------
    # The clocked region holds this rank's K steps and nothing else: barrier (all ranks start together), device
    # synchronise, clock, K steps, device synchronise, clock.  No collective sits inside it — at 20 steps the region is
    # under a millisecond and a barrier would be a tenth of it.  The per-rank times are MAX-reduced afterwards.  The block
    # is then REPEATED back to back (same K, the scenario index keeps cycling) until --timed-s of device time has been
    # clocked, and the MEDIAN block is what `value` and `ms_per_step` report: a 20-step call is not one sub-millisecond
    # sample, and the card is visibly busy for seconds to anything that samples it from outside.  `timed_repeats` says how
    # many blocks were clocked and `first_block_ms_per_step` keeps the single-sample figure.
    t_idx = legs.run_steps(eng, 0, a.warmup, a.mode or "per_step", k_steps)

    fail_rank = os.environ.get("FIVEEQ_BENCH_FAIL_RANK")        # test hook: this rank dies before the timed region
    if fail_rank is not None and int(fail_rank) == rank:
        os._exit(17)

    # Which launch form the timed region uses.  An explicit --mode is taken as given.  The default is the per-step form at
    # every N (the scaling curve must compare one launch form with itself).  For N > 1 the per-step enqueue share
    # (legs.host_enqueue: all ranks enqueuing at once) is measured first and reported (`timing.host_fallback`: a host-bound node
    # must not pass for a slow GPU); with --host-fallback a share at or above --host-share-limit switches the timed region to the
    # hipGraph replay of the same launches (same kernels, same layout, bit-identical results; 1/4 of the host time).
    k_burst = max(1, min(a.steps, 40))
    mode_requested = a.mode
    fallback = None
    if a.mode is None:
        a.mode = "per_step"
        if world > 1 or os.environ.get("FIVEEQ_BENCH_FORCE_HOST_CHECK") == "1":
            e_med, _, d_med, t_idx, _ = legs.host_enqueue(eng, ctl, "per_step", t_idx, k_burst, k_steps)
            fallback = {"per_step_host_enqueue_us_per_step": e_med * 1e6, "per_step_burst_us_per_step": d_med * 1e6,
                        "per_step_host_share": e_med / d_med, "limit": a.host_share_limit,
                        "would_switch": bool(e_med / d_med >= a.host_share_limit), "enabled": bool(a.host_fallback),
                        "switched_to_graph": False}
            if fallback["would_switch"] and a.host_fallback:
                a.mode, fallback["switched_to_graph"] = "graph", True

    def timed_block(t_from):
        """One K-step block on the wall clock: barrier, device sync, clock, K steps, drained stream, clock."""
        if a.mode == "graph":                  # instantiate the block's graphs outside the timing
            legs.prepare_graphs(eng, t_from % N_SCEN, a.steps)
        torch.cuda.synchronize(dev)
        ctl.barrier()
        torch.cuda.synchronize(dev)
        done = torch.cuda.Event()
        t0 = time.perf_counter()
        t_next = legs.run_steps(eng, t_from, a.steps, a.mode, k_steps)
        done.record()                          # on the launch stream, behind the K-th step
        while not done.query():                # spin: a blocking synchronise would add its wake-up latency to the block
            pass
        t1 = time.perf_counter()
        torch.cuda.synchronize(dev)
        return t1 - t0, t_next

    first, t_idx = timed_block(t_idx)
    first_max = ctl.max_over_ranks([first])[0]
    repeats, blocks, wall_all = 1, [first], None
    if a.timed_s > 0 and first_max < a.timed_s:
        # The repeats run BACK TO BACK: one barrier + device sync before the first, then R x K steps enqueued with a HIP
        # event on the launch stream at every block boundary, one drain at the end.  Block i = event i -> event i+1 on the
        # device's own clock, so a block holds its K steps and nothing else — no idle-stream launch latency, no host
        # wake-up — exactly what K steps cost inside a long run.  The wall clock around all R blocks is kept beside it as
        # the cross-check.  (The host runs ahead of the device until the HIP queue is full and is then paced by it.)
        # The per-step mode may run each timestep as several kernels on their own streams (engine.per_step_streams): a mark
        # is then one event PER STREAM, the blocks are not joined in between (a join is two cross-stream hops, ~20 us, that a
        # continuous run does not have), and a block lasts as long as its slowest stream takes from mark to mark.
        lanes = eng.per_step_stream_list() if a.mode == "per_step" else [torch.cuda.current_stream(dev)]

        def mark():
            evs = [torch.cuda.Event(enable_timing=True) for _ in lanes]
            for ev, lane in zip(evs, lanes):
                ev.record(lane)
            return evs

        def clock_blocks(n_blocks, t_from):
            if a.mode == "graph":
                legs.prepare_graphs(eng, t_from % N_SCEN, a.steps * min(n_blocks, -(-N_SCEN // a.steps) + 1))
            torch.cuda.synchronize(dev)
            ctl.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            marks = [mark()]
            for _ in range(n_blocks):
                t_from = legs.run_steps(eng, t_from, a.steps, a.mode, k_steps, join=False)
                marks.append(mark())
            eng.join()
            done = torch.cuda.Event()
            done.record()
            while not done.query():
                pass
            wall = time.perf_counter() - t0
            torch.cuda.synchronize(dev)
            out = [max(e0.elapsed_time(e1) for e0, e1 in zip(marks[i], marks[i + 1])) * 1e-3 for i in range(n_blocks)]
            return out, wall, t_from

        # how long IS a block inside a run?  The wall-clocked first block carries the idle-stream launch latency (10 % at 20
        # steps); a short event-timed burst sizes the main loop so that it clocks --timed-s of device time, not 10 % less
        # (always: on a box's first process the wall-clocked block has been seen 60x too long — 43 ms for 60 steps of 36 us — and a
        # main loop sized from it clocked 2 ms where 50 were asked for)
        probe, _, t_idx = clock_blocks(int(min(11, max(1, a.timed_s // max(first_max, 1e-6)))), t_idx)
        est = float(np.median(ctl.max_over_ranks(probe)))
        repeats = int(min(max(a.max_repeats, 1), -(-a.timed_s // max(est, 1e-6)))) | 1     # odd
        blocks, wall_all, t_idx = clock_blocks(repeats, t_idx)
    per_rank_ms_per_step = ctl.gather_over_ranks(float(np.median(blocks)) / a.steps * 1e3)      # each rank's own median block
    blocks = ctl.max_over_ranks(blocks)                          # per block: the slowest rank
    elapsed = float(np.median(blocks))
    value = n_total * a.steps / elapsed

    enq_med, enq_min, _, t_idx, enq_mine = legs.host_enqueue(eng, ctl, a.mode, t_idx, k_burst, k_steps)
    timing = {"timed_repeats": repeats,
              "per_rank_ms_per_step": per_rank_ms_per_step, "per_rank_host_enqueue_us": ctl.gather_over_ranks(enq_mine * 1e6),
              "per_rank_is": "rank r's own median block / K, and its own median enqueue time per step (list index = rank); "
                             "`ms_per_step` is the median over blocks of the per-block MAX over ranks",
              "block_ms_min_median_max": [min(blocks) * 1e3, elapsed * 1e3, max(blocks) * 1e3],
              "device_s_clocked": float(np.sum(blocks)),
              "first_block_ms_per_step": first_max / a.steps * 1e3,
              "first_block_is": "ONE K-step block on the wall clock (barrier, device sync, clock, K steps, drained stream, "
                                "clock; MAX over ranks): the contract's literal sample",
              "wall_ms_per_step_over_all_repeats": None if wall_all is None else
              ctl.max_over_ranks([wall_all])[0] / (a.steps * repeats) * 1e3,
              "host_enqueue_us_per_step": enq_med * 1e6, "host_enqueue_us_per_step_min": enq_min * 1e6,
              "host_share": enq_med / (elapsed / a.steps),
              "host_fallback": fallback,
              "host_enqueue_is": (f"median (and min) over 15 bursts of {k_burst} steps of the wall time this rank's thread spends "
                                  "inside engine.run -> fiveeq_run_* (enqueue only, drained device, queue never full), every "
                                  f"burst behind a barrier so that all {world} rank(s) enqueue at once; MAX over ranks; "
                                  "host_share = that / ms_per_step"),
              "clocked": ("one K-step block on the wall clock (it is longer than --timed-s)" if repeats == 1 else
                          f"{repeats} K-step blocks enqueued back to back after one barrier + device sync; block = HIP event "
                          "to HIP event on the launch stream (the slowest of the launch streams when a timestep is several "
                          "concurrent launches); MAX over ranks per block, then the median block")}
    # ---- the rows the end-of-run exchange will summarise: taken NOW, from ONE uninterrupted run (the repeated blocks
    # cycled through the scenario and overwrote stored rows with later passes; the roofline batches below overwrite more) ---
    # The WHOLE 750-step scenario, whatever K: the exchange then summarises the years 2014 / 2264 / 2514 of the run, not a
    # near-constant row 24 steps in (26 ms of device time at 1M members, outside every clocked region).
    rows, years = None, []
    if eng.T is not None:
        eng.reset_state()
        legs.run_steps(eng, 0, N_SCEN, a.mode, k_steps)
        torch.cuda.synchronize(dev)
        years = [249, 499, 749]
        rows = eng.T[years]                                     # advanced indexing: a copy

    # ---- roofline of the timed mode's kernel and the legs beside it (benchlib/legs.py; none of it inside a clocked region) ----
    mode_run = eng.resolve_mode(a.mode, k_steps)[0]              # what --mode auto resolves to on this ensemble
    fusedlike = mode_run in ("fused", "ksteps", "small")
    _, _, _, members_per_wave, _, packed = legs.kernel_tags(eng, a, n_local)
    if not fusedlike:
        roofline, k_avg, kkey = legs.per_step_roofline(eng, a, G, per_gpu, n_local, t_idx)
        if traffic_live is not None and "hbm_bytes_per_launch" in traffic_live:
            roofline["traffic_committed"] = roofline["traffic"]          # the figure of profiles/traffic.json, for comparison
            roofline["traffic"], roofline["traffic_source"] = traffic_live["hbm_bytes_per_launch"], traffic_live["source"]
            roofline["traffic_detail"] = {k: traffic_live[k] for k in ("fetch_bytes", "write_bytes", "step_dispatches", "copy_calibration", "seconds")}
        elif traffic_live is not None:
            roofline["traffic_live_error"] = traffic_live.get("error")
    else:
        roofline, k_avg, kkey, members_per_wave = legs.fused_roofline(eng, a, mode_run, k_steps, n_local)
    legs.add_valu_issue(roofline, a, kkey, k_avg, members_per_wave, packed, fusedlike)
    best_copy_gbs = legs.copy_rates(eng, dev, roofline, fusedlike)
    if not fusedlike and world == 1 and not a.no_hbm_resident and a.hbm_resident_members > 0:
        legs.hbm_resident(eng, a, p, G, dtype, dev, n_local, best_copy_gbs, roofline)

    out = {
        "metric": "ensemble_member_timesteps_per_sec", "value": value, "unit": "member-timesteps/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
        "timed_repeats": repeats, "timing": timing,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
        "data": "synthetic",
        "config": {"workload": f"{a.workload}: {desc}", "members_per_gpu": per_gpu, "members_total": n_total,
                   "gases": G, "pools": eng.pools, "scenario_steps": N_SCEN, "mode": a.mode, "mode_resolved": mode_run,
                   "mode_requested": mode_requested or "default",
                   "steps_per_launch": (roofline.get("steps_per_launch", 1)),
                   "trajectory_stored": eng.C is not None, "compensated_fp32": bool(a.compensated), "parallelism": f"member-shard x{world}",
                   "chunk_members": eng.chunk_members,
                   "collective_backend": "rccl" if backend == "nccl" else backend,
                   "control_plane": None if dist is None else "gloo over 127.0.0.1 (barriers, MAX of the clocked times)",
                   "emissions_sha256": emissions.emissions_sha256(E)[:16], "lhs_seed": params.LHS_SEED,
                   "lhs_design": "shard-computable (keyed Feistel bijection), drawn on the device",
                   "setup_s_rank0": setup_s,
                   "devices": devices},
        "roofline": legs.ordered(roofline),
    }
    if cpu is not None:
        out["cpu_baseline"] = cpu

    # ---- end-of-run exchange (the only collective that moves ensemble data; RCCL): summary statistics of T over ALL
    # members, on the rows captured above.  LAST, with the line already complete: a watchdog thread fires if the exchange has
    # not returned after --summary-watchdog-s — rank 0 then prints the line with summary.error and the process exits non-zero
    # (a fresh exit, nothing is re-executed).  Ranks other than 0 give rank 0 five more seconds before they leave, so that the
    # launcher's tear-down cannot reach rank 0 before its line is out.
    import threading
    printed, emitted = threading.Lock(), []

    def emit(summary_obj):
        """Rank 0 prints THE line, once, whoever gets here first (the main thread or the watchdog)."""
        with printed:
            if rank == 0 and not emitted:
                emitted.append(True)
                if summary_obj is not None:
                    out["summary"] = summary_obj
                print(json.dumps(out), flush=True)

    def on_timeout():
        if rank != 0:
            time.sleep(5.0)
        emit({"error": f"timeout: the summary exchange had not returned after {a.summary_watchdog_s:.0f} s"})
        print(f"rank {rank}: summary exchange timed out", file=sys.stderr, flush=True)
        os._exit(4)

    summary_error, summary_obj = None, None
    if rows is not None:
        watchdog = None
        if dist is not None:
            watchdog = threading.Timer(a.summary_watchdog_s, on_timeout)
            watchdog.daemon = True
            watchdog.start()

        def exchange(group):
            """The summary exchange over `group`, warm call timed: the `summary` object of the line (rank 0) or None."""
            stats_ = {}
            gather_summary(rows, percentiles=(5.0, 50.0, 95.0), group=group)            # warm: communicator + library set-up
            ctl.sync_all()
            ts = time.perf_counter()
            summ = gather_summary(rows, percentiles=(5.0, 50.0, 95.0), group=group, stats=stats_)
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - ts) * 1e3
            seen = {"rccl_world_size": None, "backend_seen": None}
            if dist is not None:               # as the group itself reports them, after its first collectives have run
                seen = {"rccl_world_size": dist.get_world_size(group), "backend_seen": dist.get_backend(group)}
            if rank != 0:
                return None
            return {"years": years, "gather_ms": ms, "gather_ms_is": "second (warm) call", **seen,
                    "bytes_to_root": stats_.get("bytes_to_root"), "bytes_to_root_per_rank": stats_.get("bytes_to_root_per_rank"),
                    "allreduce_bytes": stats_.get("allreduce_bytes"), "T_mean": [float(x) for x in summ["mean"]],
                    "T_p05_p50_p95": [[float(v) for v in row] for row in summ["percentiles"]]}

        try:
            if os.environ.get("FIVEEQ_BENCH_HANG_SUMMARY") == str(rank):     # test hook: this rank never enters the exchange
                time.sleep(10 * a.summary_watchdog_s + 60)
            data_group = None
            if dist is not None:               # data plane: RCCL ("nccl" on ROCm); its communicator comes up with the first collective
                data_group = dist.new_group(backend="nccl", timeout=ctl.timeout) if backend == "nccl" else dist.group.WORLD
            summary_obj = exchange(data_group)
        except Exception as exc:  # noqa: BLE001 - the measurement is complete: report the failure in the line, do not lose it
            summary_error = f"{type(exc).__name__}: {exc}"
        if dist is not None and backend == "nccl":
            # Did RCCL fail on first contact?  Every rank says so over the CONTROL plane (a rank whose peers hang inside RCCL waits
            # here until the watchdog ends the job, as before).  If any rank failed — the usual first-contact failures are
            # symmetric: every rank's communicator refuses to come up — ALL ranks run the exchange once more over the control
            # plane's gloo group (rows through host memory): the line then carries a complete summary AND the RCCL error, and the
            # job exits 0; the multi-GPU measurement is not held hostage by the one collective that is not part of it.
            try:
                errs = ctl.gather_over_ranks(summary_error)
                if any(e is not None for e in errs):
                    first = next(e for e in errs if e is not None)
                    print(f"rank {rank}: RCCL summary exchange failed ({first[:200]}): repeating it over gloo", file=sys.stderr, flush=True)
                    summary_obj = exchange(dist.group.WORLD)
                    if summary_obj is not None:
                        summary_obj["rccl_error"] = first
                        summary_obj["rccl_errors_per_rank"] = errs
                        summary_obj["fallback"] = "the exchange was repeated over the gloo control plane after RCCL failed"
                    summary_error = None
            except Exception as exc:  # noqa: BLE001
                summary_error = f"{summary_error}; gloo fallback: {type(exc).__name__}: {exc}"
        if watchdog is not None:
            watchdog.cancel()
        if summary_error is None and rank == 0:
            emit(summary_obj)
    else:
        emit(None)
    if summary_error is not None:
        # A rank that failed inside the exchange must not walk into another collective: the peers may be stuck in the one
        # it left.  Rank 0 prints its line with the error and exits; any other rank first gives rank 0 the time to reach
        # its own error or its watchdog (the launcher tears the job down as soon as one rank has exited non-zero).
        print(f"rank {rank}: summary exchange failed: {summary_error}", file=sys.stderr, flush=True)
        if rank != 0:
            time.sleep(a.summary_watchdog_s + 5.0 if world > 1 else 0.0)
        emit({"error": summary_error})
        os._exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# =====================================================================================================================
# Below: what runs BEFORE main() touches a GPU — the plain `--gpus N` self-launch and the CPU-baseline leg (the only code of this
# file that imports oracle/, and only as the thing timed on the host cores: baseline only).
# =====================================================================================================================
def _self_launch():
    """Plain `python bench.py --gpus N ...` with N > 1 and no launcher environment (the way the driver calls --gpus 1):
    this process — BEFORE it imports torch or touches a GPU — starts
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same args>
    as a CHILD process (its own session, so that a timeout can end the whole group), relays the child's stdout (rank 0's
    one JSON line) and stderr, and exits with the child's code.  It never initialises the GPU itself: no exec from a
    process that has.  The explicit launcher form (WORLD_SIZE / RANK in the environment) does not come through here."""
    if "WORLD_SIZE" in os.environ or "RANK" in os.environ or "--numpy-worker" in sys.argv:
        return
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--launch-timeout-s", type=float, default=570.0)
    known, _ = ap.parse_known_args()
    if known.gpus <= 1:
        return
    import signal
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={known.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this host driver
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)      # stdout / stderr inherited: the line passes through
    try:
        rc = proc.wait(timeout=known.launch_timeout_s)
    except subprocess.TimeoutExpired:
        print(f"bench.py: the {known.gpus}-rank child job exceeded --launch-timeout-s {known.launch_timeout_s:.0f}: "
              "ending its process group", file=sys.stderr, flush=True)
        try:
            os.killpg(proc.pid, signal.SIGTERM)
            proc.wait(timeout=15)
        except Exception:  # noqa: BLE001
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except Exception:  # noqa: BLE001
                pass
        rc = 124
    except KeyboardInterrupt:
        os.killpg(proc.pid, signal.SIGTERM)
        rc = 130
    sys.exit(rc)


def _numpy_worker(argv):
    """`bench.py --numpy-worker kind G members steps lo hi`: one process of the `numpy_nproc` CPU-baseline leg — advances
    members [lo, hi) of the `members`-member sample with the NumPy oracle and prints its compute seconds.  No torch, no GPU."""
    kind, G, n, steps, lo, hi = argv[0], int(argv[1]), int(argv[2]), int(argv[3]), int(argv[4]), int(argv[5])
    from fiveeqscm_amd import emissions, params
    from oracle import fiveeq_oracle as npo
    p = params.sample_ensemble_shard(params.default_params(kind), n, lo, hi)
    E = emissions.rcp_like_emissions(steps, G)
    t0 = time.perf_counter()
    npo.run(E, p, hi - lo)
    print(f"numpy-worker {time.perf_counter() - t0:.6f}", flush=True)


def _usable_cores():
    """Host threads this process may really use: affinity mask, capped by the cgroup CPU quota
    (a GPU box exposes all hardware threads of the node but grants one GPU's share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:  # noqa: BLE001
        pass
    return n


def numpy_nproc(kind, G, n_members, n_steps, n_proc):
    """SURVEY 8d's second CPU leg: `n_proc` independent NumPy processes (one per usable core), each on a contiguous shard
    of `n_members` members; reported on the compute wall time of the SLOWEST process (imports excluded).  Child processes
    (fork + exec of a fresh interpreter that never touches the GPU)."""
    import subprocess
    bounds = [(r * n_members // n_proc, (r + 1) * n_members // n_proc) for r in range(n_proc)]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--numpy-worker", kind, str(G), str(n_members),
                               str(n_steps), str(lo_), str(hi_)], stdout=subprocess.PIPE, text=True, env=env)
             for lo_, hi_ in bounds]
    times = []
    for pr in procs:
        out, _ = pr.communicate(timeout=600)
        line = [ln for ln in out.splitlines() if ln.startswith("numpy-worker")]
        if pr.returncode != 0 or not line:
            return {"error": f"worker failed (rc {pr.returncode})"}
        times.append(float(line[0].split()[1]))
    return {"value": n_members * n_steps / max(times), "processes": n_proc, "gases": G,
            "sample": f"{n_members} members x {n_steps} steps over {n_proc} NumPy processes (one per usable core, contiguous "
                      f"member shards), compute time of the slowest {max(times):.2f} s, fastest {min(times):.2f} s"}


def cpu_baseline(kind, G, n_sample, n_steps, numpy_legs=False):
    """Time the plain-C oracle (oracle/fiveeq_oracle.c, OpenMP over members) and the NumPy oracle
    (one core) on a bounded sample of the same workload.  Baseline only.  The thread count is the
    fastest of {usable cores, 64, 32, 16} on a small probe, and is what `cores` reports."""
    from fiveeqscm_amd import emissions, params
    from oracle import c_oracle, fiveeq_oracle as npo
    p = params.sample_ensemble(params.default_params(kind), n_sample)
    E = emissions.rcp_like_emissions(n_steps, G)

    def sub(n):
        ps = dict(p)
        for k in ("r0", "rC", "rT", "q"):
            ps[k] = p[k][:, :n]
        return ps

    usable = _usable_cores()
    n_probe = min(n_sample, 8192)
    best, cores = None, 1
    for th in sorted({usable, min(usable, 64), min(usable, 32), min(usable, 16)}):
        c_oracle.run(E, sub(n_probe), n_probe, n_threads=th, keep=())          # warm-up (threads, page-in)
        t0 = time.perf_counter()
        c_oracle.run(E, sub(n_probe), n_probe, n_threads=th, keep=())
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, th
    t0 = time.perf_counter()
    c_oracle.run(E, p, n_sample, n_threads=cores, keep=())      # state only: no 9 GB host trajectory
    dt_c = time.perf_counter() - t0
    n_np = min(n_sample, 10_000)
    t0 = time.perf_counter()
    npo.run(E, sub(n_np), n_np)
    dt_np = time.perf_counter() - t0
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            model = next(line.split(":", 1)[1].strip() for line in fh if line.startswith("model name"))
    except Exception:  # noqa: BLE001
        pass
    extra = {"numpy_nproc": numpy_nproc(kind, G, 20_000 * usable, n_steps, usable)}
    if numpy_legs:                      # SURVEY 8d in full: N = 1e5 on one core and on every core, CO2-only and multi-gas
        for kind_, G_ in (("co2", 1), ("multigas", 3)):
            pk = params.sample_ensemble(params.default_params(kind_), 100_000)
            Ek = emissions.rcp_like_emissions(n_steps, G_)
            t0 = time.perf_counter()
            npo.run(Ek, pk, 100_000)
            dt_k = time.perf_counter() - t0
            extra[f"numpy_1core_1e5_{kind_}"] = {"value": 100_000 * n_steps / dt_k,
                                                 "sample": f"100000 members x {n_steps} steps, {G_} gas(es), {dt_k:.2f} s"}
            extra[f"numpy_nproc_1e5_{kind_}"] = numpy_nproc(kind_, G_, 100_000, n_steps, usable)
    c_port = {"value": n_sample * n_steps / dt_c, "cores": cores,
              "sample": f"{n_sample} members x {n_steps} steps, {G} gas(es), fp64, oracle/fiveeq_oracle.c "
                        f"(gcc -O2 -fopenmp, {cores} threads of {usable} usable; final state kept, trajectories not stored), "
                        f"{dt_c:.2f} s"}
    # `value` is the STRONGER of the two whole-box CPU figures: the NumPy oracle as one process per usable core on member
    # shards (what north_star names: "the reference NumPy loop timed on the same box's host cores") or the C port under
    # OpenMP.  NumPy's vectorised transcendentals beat the scalar-libm C loop per core, so it is usually the former.
    nn = extra["numpy_nproc"]
    best_numpy = "value" in nn and nn["value"] > c_port["value"]
    return {
        **extra,
        "c_port_openmp": c_port,
        "note": "baseline only.  value = the stronger of (NumPy oracle, one process per usable core on contiguous member "
                "shards) and (C port of the oracle, OpenMP over members); both are reported.",
        "value": nn["value"] if best_numpy else c_port["value"], "unit": "member-timesteps/s",
        "cores": nn["processes"] if best_numpy else cores, "kind": "port",
        "sample": (f"oracle/fiveeq_oracle.py (numpy {np.__version__}): " + nn["sample"]) if best_numpy else c_port["sample"],
        "cpu_model": model,
        "numpy_1core": {"value": n_np * n_steps / dt_np, "sample": f"{n_np} members x {n_steps} steps, "
                        f"oracle/fiveeq_oracle.py, numpy {np.__version__}, {dt_np:.2f} s"},
    }


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--numpy-worker":
        _numpy_worker(sys.argv[2:])
        sys.exit(0)
    _self_launch()
    main()
