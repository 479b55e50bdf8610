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
enqueuing at once.  `roofline` prices the kernel of the chosen --mode: the per-step kernel against the
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
ensemble data — goes over RCCL, LAST, with the line already complete and a watchdog thread beside it: a failed or hung
exchange costs the line its `summary` (-> {"error": ...}, non-zero exit), never its measurement.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


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


if len(sys.argv) > 1 and sys.argv[1] == "--numpy-worker":
    _numpy_worker(sys.argv[2:])
    sys.exit(0)



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


if __name__ == "__main__":
    _self_launch()

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# Vector issue peak: 256 CUs x 4 SIMDs at 2.4 GHz.  A wave64 fp64 instruction and a PACKED fp32 instruction (two fp32 ops
# per lane) hold their SIMD for 4 cycles — that is what the datasheet's 78.6 TFLOP/s fp64 (v_fma_f64) and 157.3 TFLOP/s fp32
# (v_pk_fma_f32) are; a scalar fp32 or integer instruction nominally holds it for 2.  The roofline of the VALU-bound
# kernels is the NOMINAL ISSUE TIME of their measured instruction stream (SQ counters: instructions per wave-step, and for
# the fp32 kernels the packed share of them) divided by the measured time.  What this chip SUSTAINS on pure streams of one
# instruction kind (8 waves/SIMD, tools/microbench/valu_rates.hip, profiles/r03/valu_rates_microbench.txt) is quoted beside
# it: the clock it holds under a dense VALU stream is 1.96-2.03 GHz (SQ counters), not 2.4, and nothing reaches nominal.
SIMDS = 1024
CLOCK_HZ = 2.4e9
VALU_CYCLES_PER_INSTR = {"f64": 4.0, "f32": 4.0}
VALU_SUSTAINED_CYCLES = {"f64": {"v_fma_f64": 5.52, "v_add_f64": 4.93, "v_rcp_f64": 17.45},
                         "f32": {"v_pk_fma_f32": 5.18, "v_pk_mul_f32": 5.00, "v_pk_add_f32": 4.81, "v_fma_f32": 3.58,
                                 "v_rcp_f32": 8.41}}

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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hbm-resident", action="store_true", help="skip the beyond-Infinity-Cache roofline leg")
    ap.add_argument("--hbm-resident-members", type=int, default=8_000_000)
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


def run_steps(eng, t0, k, mode, k_steps, join=True):
    """Advance k model timesteps starting at scenario index t0 (cycling); returns the next index."""
    n = eng.n_steps
    t = t0 % n
    while k > 0:
        seg = min(k, n - t)
        eng.run(t, t + seg, mode=mode, k_steps=k_steps, join=join)
        k -= seg
        t = (t + seg) % n
    return t


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


def event_timed(eng, launch, t_idx, n_scen, span, batches, lanes=None):
    """Average duration of one `launch(t, t + span)` (HIP events on the launch stream(s), queue kept busy ahead of the
    first event): list of seconds per batch.  `lanes`: the streams the launches run on when a timestep is several
    concurrent launches (engine.per_step_stream_list()); `launch` must then not join them (run(..., join=False)): an event
    is recorded on every lane and a batch lasts as long as its slowest lane takes from mark to mark."""
    samples = []
    lead = min(5, max(1, span))
    lanes = lanes or [torch.cuda.current_stream()]

    def mark():
        evs = [torch.cuda.Event(enable_timing=True) for _ in lanes]
        for ev, lane in zip(evs, lanes):
            ev.record(lane)
        return evs

    for i in range(max(batches, 1)):
        t = (t_idx + i * span) % max(1, n_scen - span - lead)          # t + lead + span <= n_scen always
        launch(t, t + lead)                                             # keep the queues busy ahead of the first events
        m0 = mark()
        launch(t + lead, t + lead + span)
        m1 = mark()
        eng.join()
        for ev in m1:
            ev.synchronize()
        samples.append(max(a_.elapsed_time(b_) for a_, b_ in zip(m0, m1)) * 1e-3)
    return np.array(samples)


def load_profile_json(name, key):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    try:
        with open(path) as fh:
            return json.load(fh).get(key)
    except Exception:  # noqa: BLE001
        return None


class _stdout_to_stderr:
    """File descriptor 1 points at stderr inside the block (native libraries that print to stdout do not go through sys.stdout)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:       # only reachable when bench.main() is called from other code: the script self-launches
            sys.exit(f"--gpus {a.gpus} needs WORLD_SIZE / RANK / LOCAL_RANK from a launcher (run bench.py as a script)")
        a.gpus = world
    kind, G, per_gpu, desc = WORKLOADS[a.workload]
    per_gpu = a.members or per_gpu
    n_total = per_gpu * world
    n_scen = 750

    # ---- CPU baseline FIRST (rank 0 of a one-GPU run only): the GPU has not been touched yet, so everything after this leg
    # is one contiguous window of GPU work for whoever samples the card from outside ------------------------------------
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(kind, G, a.cpu_sample_members, n_scen, numpy_legs=a.numpy_baseline)

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
    dist, data_group, tmo = None, None, None
    # FIVEEQ_BENCH_FORCE_DIST=1: build the process groups and run every collective of the N > 1 path (barriers, the MAX
    # of the block times, the summary exchange) in a ONE-rank job too — RCCL first contact for this file on a one-GPU box.
    force_dist = world == 1 and os.environ.get("FIVEEQ_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        # TWO process groups.  CONTROL plane (default group): gloo over 127.0.0.1 — the barriers around the timed region and
        # the MAX of the block times, host scalars only.  DATA plane: RCCL ("nccl" on ROCm) — the end-of-run summary
        # exchange, the only collective that moves ensemble data; its communicator is created by its first collective, which
        # happens AFTER the measurement and under a watchdog.  Whatever RCCL does on first contact across xGMI (an exception,
        # a hang) can therefore cost the line its `summary`, never its measurement.  Every group has a timeout.
        from datetime import timedelta

        import torch.distributed as dist
        if force_dist:
            for key, val in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29513"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                os.environ.setdefault(key, val)
            from fiveeqscm_amd.distributed import force_collectives
            force_collectives(True)
        tmo = timedelta(seconds=a.dist_timeout_s)
        with _stdout_to_stderr():          # gloo announces its connections on STDOUT: the one line must stay the only one
            dist.init_process_group("gloo", timeout=tmo)
            dist.barrier()
        # (the RCCL group itself is created where it is first used: inside the watchdog-protected summary section below)

    from fiveeqscm_amd import emissions, params
    from fiveeqscm_amd.distributed import gather_summary, shard_bounds
    from fiveeqscm_amd.engine import EnsembleEngine

    dtype = torch.float64 if a.dtype == "f64" else torch.float32
    k_steps = a.k_steps or None

    # This rank's contiguous shard [lo, hi) of ONE Latin hypercube over all members, drawn on this
    # rank's GPU: O(shard) work and memory whatever the world size, identical design for any world size.
    t_setup = time.perf_counter()
    lo, hi = shard_bounds(n_total, rank, world)
    p = params.sample_ensemble_shard(params.default_params(kind), n_total, lo, hi, device=dev, dtype=dtype)
    E = emissions.rcp_like_emissions(n_scen, G)
    eng = EnsembleEngine(p, hi - lo, E, dtype=dtype, device=dev, store_trajectory=not a.no_trajectory)
    torch.cuda.synchronize(dev)
    setup_s = time.perf_counter() - t_setup

    def barrier():
        if dist is not None:
            dist.barrier()                                       # control plane (gloo)

    def sync_all():
        torch.cuda.synchronize(dev)
        barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(values):
        """Element-wise MAX over the ranks of a list of floats (a control-plane collective, never inside a clocked region)."""
        if dist is None:
            return [float(v) for v in values]
        tt = torch.tensor(values, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return [float(v) for v in tt.tolist()]

    def gather_over_ranks(obj):
        """Every rank's (small, picklable) `obj` as a list indexed by rank, on every rank — over the control plane."""
        if dist is None:
            return [obj]
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    props = torch.cuda.get_device_properties(dev)
    devices = gather_over_ranks({"rank": rank, "local_rank": local_rank, "device_index": dev_index, "name": props.name,
                                 "pci_bus_id": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                                                   getattr(props, "pci_device_id", 0)),
                                 "uuid": str(getattr(props, "uuid", "")), "pid": os.getpid(),
                                 "visible_devices": torch.cuda.device_count()})

    # ---- device spin-up (not model work): the GPU idles at its lowest clock during host set-up;
    # ~30 ms of a plain copy kernel brings it back so that a small W is enough ----
    spin_src = torch.empty(1 << 25, dtype=torch.float64, device=dev).normal_()
    spin_dst = torch.empty_like(spin_src)
    for _ in range(256):
        eng.lib.fiveeq_stream_copy_f64(spin_src.numel(), ctypes.c_void_p(spin_src.data_ptr()),
                                       ctypes.c_void_p(spin_dst.data_ptr()), eng._stream())
    torch.cuda.synchronize(dev)
    del spin_src, spin_dst

    # ---- warm-up, then EXACTLY K timed steps ------------------------------------------------------
    # The clocked region holds this rank's K steps and nothing else: barrier (all ranks start together), device
    # synchronise, clock, K steps, device synchronise, clock.  No collective sits inside it — at 20 steps the region is
    # under a millisecond and a barrier would be a tenth of it.  The per-rank times are MAX-reduced afterwards.  The block
    # is then REPEATED back to back (same K, the scenario index keeps cycling) until --timed-s of device time has been
    # clocked, and the MEDIAN block is what `value` and `ms_per_step` report: a 20-step call is not one sub-millisecond
    # sample, and the card is visibly busy for seconds to anything that samples it from outside.  `timed_repeats` says how
    # many blocks were clocked and `first_block_ms_per_step` keeps the single-sample figure.
    t_idx = run_steps(eng, 0, a.warmup, a.mode or "per_step", k_steps)

    def prepare_graphs(t_from, k):
        while k > 0:
            seg = min(k, n_scen - t_from)
            eng.prepare_graph(t_from, t_from + seg)
            k -= seg
            t_from = (t_from + seg) % n_scen

    fail_rank = os.environ.get("FIVEEQ_BENCH_FAIL_RANK")        # test hook: this rank dies before the timed region
    if fail_rank is not None and int(fail_rank) == rank:
        os._exit(17)

    # ---- the HOST side of a step: how long this rank's CPU thread needs to ENQUEUE one timestep (Python + ctypes + the
    # hipLaunchKernel calls inside fiveeq_run_*), measured on a drained device with a short burst so that the HIP queue never
    # fills (a full queue blocks the caller: that would clock the device, not the host).  Every sample starts behind a
    # barrier, so with N ranks all N host threads enqueue AT THE SAME TIME — the contention an 8-GPU node's host side sees.
    # host_share = enqueue time / step time: the fraction of a step the host thread is busy; < 1 means the device, not the
    # host, paces the run (north_star's >= 7x at 8 GPUs needs this to stay well below 1 with 8 ranks enqueuing at once).
    k_burst = max(1, min(a.steps, 40))

    def host_enqueue(mode, t_from):
        """(median, min) seconds per step over 15 bursts, MAX over ranks; the device time per step of the same bursts
        (HIP events, MAX over ranks, median); the next scenario index."""
        enq, dev_t = [], []
        for _ in range(15):
            if mode == "graph":
                prepare_graphs(t_from % n_scen, k_burst)
            sync_all()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            t0 = time.perf_counter()
            t_from = run_steps(eng, t_from, k_burst, mode, k_steps, join=False)
            enq.append((time.perf_counter() - t0) / k_burst)
            eng.join()
            e1.record()
            e1.synchronize()
            dev_t.append(e0.elapsed_time(e1) * 1e-3 / k_burst)
        torch.cuda.synchronize(dev)
        host_enqueue.mine = float(np.median(enq))            # this rank's own figure (gathered per rank for the line)
        med, mn, dmed = max_over_ranks([float(np.median(enq)), float(np.min(enq)), float(np.median(dev_t))])
        return med, mn, dmed, t_from

    # Which launch form the timed region uses.  An explicit --mode is taken as given.  The default is the per-step form at
    # every N (the scaling curve must compare one launch form with itself).  For N > 1 the per-step enqueue share is measured
    # first with all ranks enqueuing at once and reported (`timing.host_fallback`: a host-bound node must not pass for a slow
    # GPU); with --host-fallback a share at or above --host-share-limit switches the timed region to the hipGraph replay of the
    # same launches (same kernels in the same layout, bit-identical results; 1/4 of the host time).  `config.mode` says what ran.
    mode_requested = a.mode
    fallback = None
    if a.mode is None:
        a.mode = "per_step"
        if world > 1 or os.environ.get("FIVEEQ_BENCH_FORCE_HOST_CHECK") == "1":
            e_med, _, d_med, t_idx = host_enqueue("per_step", t_idx)
            fallback = {"per_step_host_enqueue_us_per_step": e_med * 1e6, "per_step_burst_us_per_step": d_med * 1e6,
                        "per_step_host_share": e_med / d_med, "limit": a.host_share_limit,
                        "would_switch": bool(e_med / d_med >= a.host_share_limit), "enabled": bool(a.host_fallback),
                        "switched_to_graph": False}
            if fallback["would_switch"] and a.host_fallback:
                a.mode, fallback["switched_to_graph"] = "graph", True

    def timed_block(t_from):
        """One K-step block on the wall clock: barrier, device sync, clock, K steps, drained stream, clock.  Also returns
        the host's share of it: the time until the last launch call had returned."""
        if a.mode == "graph":                  # instantiate the block's graphs outside the timing
            prepare_graphs(t_from % n_scen, a.steps)
        torch.cuda.synchronize(dev)
        barrier()
        torch.cuda.synchronize(dev)
        done = torch.cuda.Event()
        t0 = time.perf_counter()
        t_next = run_steps(eng, t_from, a.steps, a.mode, k_steps)
        t_enq = time.perf_counter()
        done.record()                          # on the launch stream, behind the K-th step
        while not done.query():                # spin: a blocking synchronise would add its wake-up latency to the block
            pass
        t1 = time.perf_counter()
        torch.cuda.synchronize(dev)
        return t1 - t0, t_next, t_enq - t0

    first, t_idx, _ = timed_block(t_idx)
    first_max = max_over_ranks([first])[0]
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
                prepare_graphs(t_from % n_scen, a.steps * min(n_blocks, -(-n_scen // a.steps) + 1))
            torch.cuda.synchronize(dev)
            barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            marks = [mark()]
            for _ in range(n_blocks):
                t_from = run_steps(eng, t_from, a.steps, a.mode, k_steps, join=False)
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
        est = first_max
        if 12 * first_max < a.timed_s:
            probe, _, t_idx = clock_blocks(11, t_idx)
            est = float(np.median(max_over_ranks(probe)))
        repeats = int(min(max(a.max_repeats, 1), -(-a.timed_s // max(est, 1e-6)))) | 1     # odd
        blocks, wall_all, t_idx = clock_blocks(repeats, t_idx)
    per_rank_ms_per_step = gather_over_ranks(float(np.median(blocks)) / a.steps * 1e3)      # each rank's own median block
    blocks = max_over_ranks(blocks)                              # per block: the slowest rank
    elapsed = float(np.median(blocks))
    value = n_total * a.steps / elapsed

    enq_med, enq_min, _, t_idx = host_enqueue(a.mode, t_idx)
    per_rank_enq_us = gather_over_ranks(host_enqueue.mine * 1e6)
    timing = {"timed_repeats": repeats,
              "per_rank_ms_per_step": per_rank_ms_per_step, "per_rank_host_enqueue_us": per_rank_enq_us,
              "per_rank_is": "rank r's own median block / K, and its own median enqueue time per step (list index = rank); "
                             "`ms_per_step` is the median over blocks of the per-block MAX over ranks", "block_ms_min_median_max": [min(blocks) * 1e3, elapsed * 1e3, max(blocks) * 1e3],
              "device_s_clocked": float(np.sum(blocks)),
              "first_block_ms_per_step": first_max / a.steps * 1e3,
              "first_block_is": "ONE K-step block on the wall clock (barrier, device sync, clock, K steps, drained stream, "
                                "clock; MAX over ranks): the contract's literal sample",
              "wall_ms_per_step_over_all_repeats": None if wall_all is None else
              max_over_ranks([wall_all])[0] / (a.steps * repeats) * 1e3,
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
        run_steps(eng, 0, n_scen, a.mode, k_steps)
        torch.cuda.synchronize(dev)
        years = [249, 499, 749]
        rows = eng.T[years]                                     # advanced indexing: a copy

    # ---- roofline: per-launch duration of the timed mode's kernel, HIP events on the launch stream ----
    # The engine launches on torch's current stream, so torch.cuda.Event (hipEvent) brackets the launches.  Each
    # sample = one batch of launches enqueued back-to-back from C between two events: the queue stays full, so the
    # quotient is the kernel's duration plus the ~1-2 us dependent-launch boundary (a single bracketed launch would
    # add the ~10 us idle-stream launch latency instead and overstate the kernel).
    n_local = hi - lo
    valu_peak = SIMDS * CLOCK_HZ / VALU_CYCLES_PER_INSTR[a.dtype]
    tname = "double" if a.dtype == "f64" else "float"
    wbytes = 8 if a.dtype == "f64" else 4
    # fp32 runs the packed kernels (two members per lane) whenever the rows allow 8-byte accesses: even members per launch
    packed = a.dtype == "f32" and n_local % 2 == 0 and (eng.chunk_members % 2 == 0)
    lname = "float2 (two members per lane)" if packed else tname
    vtag = a.dtype + ("x2" if packed else "")
    members_per_wave = 128 if packed else 64
    pools3 = ",".join(str(x) for x in (eng.pools + [0, 0])[:3])
    mode_run = eng.resolve_mode(a.mode, k_steps)[0]              # what --mode auto resolves to on this ensemble
    fusedlike = mode_run in ("fused", "ksteps", "small")
    if not fusedlike:
        A = eng.bytes_per_member_step("per_step")
        # One timestep = n_seq member chunks one after the other (chunk-major schedule of large ensembles) x `conc` parts of
        # a chunk side by side on their own HIP streams (engine.per_step_streams; graph replay uses the same layout).  The
        # launches of the `conc` parts overlap fully — each stream issues its next kernel the moment its last one ends — so
        # the period of a chunk's step is also what each of those kernels lasts: `avg_launch_us` below is that period, the
        # figure rocprofv3 --kernel-trace reports as the kernel's average duration, and the chip moves `conc` launches'
        # bytes in it.
        layout = eng.per_step_launches()
        conc = 1 + max(si for _, _, si in layout)
        n_seq = len(layout) // conc
        n_launch = len(layout)
        members_per_launch = n_local / n_launch
        per_batch = 100
        # (--mode graph replays the same kernels in the same layout: their duration is measured on eagerly enqueued launches,
        # so that no graph capture falls between two marks)
        samples = event_timed(eng, lambda t0_, t1_: eng.run(t0_, t1_, mode="per_step", join=False), t_idx, n_scen, per_batch,
                              a.kernel_batches, lanes=eng.per_step_stream_list())
        samples = samples / (per_batch * n_seq)
        k_avg = float(samples.mean())
        achieved = A * members_per_launch * conc / k_avg / 1e9
        tkey = f"{a.workload}:{a.dtype}:{per_gpu}"
        traffic = (load_profile_json("traffic.json", tkey) or {}).get("hbm_bytes_per_launch")
        resident = wbytes * (eng.sum_pools + 2 + 3 * G + 2) * members_per_launch * conc   # state + parameter rows of a chunk
        if resident <= 0.8 * (256 << 20):
            note = (f"achieved = algorithmic bytes / kernel time.  At {int(members_per_launch * conc)} members per step the "
                    f"{resident / 1e6:.0f} MB of state + parameters stay in the 256 MiB Infinity Cache between launches, so "
                    "this is HBM-peak-priced algorithmic traffic, not bytes that crossed HBM; `hbm_resident` is the same "
                    "kernel with nothing cached.")
        else:
            note = (f"achieved = algorithmic bytes / kernel time.  {resident / 1e6:.0f} MB of state + parameters per launch "
                    "against a 256 MiB Infinity Cache: most of these bytes cross HBM every launch"
                    + (" (chunk-major schedule: one member chunk at a time stays cached between its launches)."
                       if n_seq > 1 else "."))
        if conc > 1:
            note += (f"  {conc} launches of {int(members_per_launch)} members each run side by side on their own streams: "
                     f"achieved = {conc} x algorithmic bytes per launch / the launch duration.")
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": (f"profiles/traffic.json[{tkey}]: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                       "tools/collect_profiles.sh, calibrated on a known copy in the same pass — a committed "
                                       "measurement of this kernel and size, NOT re-measured by this run")
                    if traffic is not None else None,
                    "kernel": f"fiveeq::step_kernel<{lname},{pools3}>",
                    "algorithmic_bytes_per_member_step": A, "members_per_launch": members_per_launch,
                    "concurrent_launches": conc, "sequential_chunks_per_step": n_seq,
                    "algorithmic_bytes_per_launch": A * members_per_launch,
                    "achieved_per_launch": A * members_per_launch / k_avg / 1e9,
                    "avg_launch_us": k_avg * 1e6, "min_launch_us": float(samples.min()) * 1e6,
                    "launches_timed": int(samples.size) * per_batch * n_launch, "note": note}
        # north_star's literal launch shape — ONE kernel per timestep, one stream — beside the default above (from ~0.5M members
        # the engine runs a timestep as two launches over member halves on two streams: a measured -5...-8 %, same bits)
        if conc > 1 and n_seq == 1:
            eng.join()
            saved_streams, eng.per_step_streams = eng.per_step_streams, 1
            one = event_timed(eng, lambda t0_, t1_: eng.run(t0_, t1_, mode="per_step", join=False), t_idx, n_scen, per_batch,
                              a.kernel_batches, lanes=eng.per_step_stream_list()) / per_batch
            eng.per_step_streams = saved_streams
            k_one = float(one.mean())
        else:
            k_one = k_avg                                       # the default already is one launch per timestep (and chunk)
        ach_one = A * members_per_launch * conc / k_one / 1e9
        roofline["single_launch"] = {"avg_launch_us": k_one * 1e6, "achieved": ach_one, "frac": ach_one / HBM_PEAK_GBS,
                                     "members_per_launch": members_per_launch * conc,
                                     "is": "the same kernel as ONE launch per timestep on one stream (per_step_streams=1), "
                                           "100-launch HIP-event batches like avg_launch_us"}
        roofline["single_launch_avg_us"], roofline["single_launch_achieved"] = k_one * 1e6, ach_one
        roofline["single_launch_frac"] = ach_one / HBM_PEAK_GBS
        kkey = f"step:{vtag}:{pools3}"
    else:
        # the time-fused family: one launch covers `span` steps; price it per step with its own A
        if mode_run == "small":
            lpm = eng.small_form()
            span, kname, mode_t = n_scen, "small_kernel", "small"
            single = len(eng.pools) == 1
            lname = f"{tname},{eng.pools[0]},{lpm}" if single else f"{tname},{pools3}"
            vtag, members_per_wave = a.dtype, 64 // lpm                  # (never packed)
        elif mode_run == "fused":
            span, kname, mode_t = eng.fused_span_steps(n_scen), "fused_kernel", "fused"     # the engine relaunches small ensembles
        else:
            span = k_steps or eng.auto_k_steps()
            kname, mode_t = "fused_kernel", "ksteps"
        A = eng.bytes_per_member_step(mode_t, span if mode_t == "ksteps" else None)
        # Timed the way the timed region runs it: whole scenario passes from the initial state (HIP events on the
        # launch stream around each pass; the launches of a pass are enqueued back-to-back from C).
        samples = []
        for _ in range(max(a.kernel_batches, 2)):
            eng.reset_state()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.run(0, n_scen, mode=mode_t, k_steps=span if mode_t == "ksteps" else None)
            e1.record()
            e1.synchronize()
            samples.append(e0.elapsed_time(e1) * 1e-3 / n_scen)
        samples = np.array(samples[1:])                         # the first pass re-warms
        reps = -(-n_scen // span)
        k_avg = float(samples.mean())                           # seconds per model step inside the kernel
        achieved = A * n_local / k_avg / 1e9
        kernel_name = ((f"fiveeq::small_kernel<{lname},false>" if single else f"fiveeq::small_multi_kernel<{lname},false>")
                       if kname == "small_kernel" else f"fiveeq::{kname}<{lname},{pools3}>")
        roofline = {"bound": "fp64-valu" if a.dtype == "f64" else "fp32-valu", "unit": "wave-instr/s",
                    "achieved": None, "peak": valu_peak, "frac": None, "traffic": None,
                    "kernel": kernel_name, "steps_per_launch": span,
                    "algorithmic_bytes_per_member_step": A, "members_per_launch": n_local,
                    "hbm_GBs_of_algorithmic_bytes": achieved, "hbm_frac": achieved / HBM_PEAK_GBS,
                    "avg_step_us_in_kernel": k_avg * 1e6, "launches_timed": int(samples.size) * reps,
                    "timed_as": f"{samples.size} whole {n_scen}-step scenario passes from the initial state",
                    "note": "time-fused family: state stays in registers, the kernel is bound by VALU issue, not HBM; "
                            "frac = VALU wave-instructions per second / (1024 SIMDs x 2.4 GHz / cycles per instruction)."}
        if kname == "small_kernel":
            waves_ = -(-n_local // members_per_wave)
            roofline["lanes_per_member"], roofline["waves"] = lpm, waves_
            roofline["note"] = (f"small-ensemble kernel: {lpm} lane(s) per member, {waves_} waves for {SIMDS} SIMDs — a wave alone on "
                                "its SIMD issues one vector instruction per ~3.7-4.2 ns whatever the instruction, so the run is bound "
                                "by the instructions ONE wave issues per step (valu_issue.valu_wave_instr_per_wave_step), not by the "
                                "chip's VALU peak: frac prices the waves that exist against all 1024 SIMDs at nominal issue.")
        kkey = (f"small:{vtag}:{pools3}:{lpm}" if kname == "small_kernel" else f"fused:{vtag}:{pools3}")
    # VALU issue: instructions per wave-step from the committed SQ-counter pass (profiles/valu.json, produced by
    # tools/collect_profiles.sh with rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES ...), times the waves this bench ran
    valu = load_profile_json("valu.json", kkey)
    if valu:
        waves = -(-int(roofline["members_per_launch"] * roofline.get("concurrent_launches", 1)) // members_per_wave)
        rate = valu["valu_per_wave_step"] * waves / k_avg
        # nominal issue time of the stream: 4 cycles per fp64 or packed-fp32 wave-instruction, 2 per scalar fp32 / integer one
        # (the datasheet's 78.6 / 157.3 TFLOP/s are v_fma_f64 and v_pk_fma_f32 at 4 cycles); the packed share of an fp32
        # stream comes from the SQ_INSTS_VALU_FLOPS_FP32 pass (profiles/valu.json "packed_per_wave_step")
        n_valu = valu["valu_per_wave_step"]
        n_slow = n_valu if a.dtype == "f64" else valu.get("packed_per_wave_step", n_valu if packed else 0.0)
        nominal_cycles = 4.0 * n_slow + 2.0 * (n_valu - n_slow)
        nominal_s = nominal_cycles * waves / (SIMDS * CLOCK_HZ)
        issue = {"valu_wave_instr_per_wave_step": n_valu, "members_per_wave": members_per_wave,
                 "four_cycle_instr_per_wave_step": n_slow, "nominal_issue_cycles_per_wave_step": nominal_cycles,
                 "wave_instr_per_s": rate, "peak_wave_instr_per_s": valu_peak * (4.0 * n_valu / nominal_cycles),
                 "frac": nominal_s / k_avg,
                 "peak_def": "frac = nominal issue time of the kernel's VALU stream / measured time, on 1024 SIMDs x 2.4 GHz with "
                             "4 cycles per fp64 or packed-fp32 wave-instruction (the datasheet's 78.6 TFLOP/s v_fma_f64 and "
                             "157.3 TFLOP/s v_pk_fma_f32) and 2 per scalar fp32 / integer one",
                 "measured_sustained_cycles_per_instr": VALU_SUSTAINED_CYCLES[a.dtype],
                 "measured_sustained_source": "profiles/r03/valu_rates_microbench.txt (pure streams, 8 waves/SIMD)",
                 "clock_GHz_under_load": valu.get("clock_GHz_under_load"),
                 "source": f"profiles/valu.json[{kkey}] (rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES ..., committed)"}
        if fusedlike:
            roofline["achieved"], roofline["frac"], roofline["peak"] = rate, issue["frac"], issue["peak_wave_instr_per_s"]
        roofline["fp64_issue_frac" if a.dtype == "f64" else "fp32_issue_frac"] = issue["frac"]
        roofline["valu_issue"] = issue

    # achievable copy bandwidth on this box, same access shape (8 B/lane), buffers beyond the 256 MiB L3
    n_copy = 1 << 27                                        # 1 GiB read + 1 GiB written per launch
    src = torch.empty(n_copy, dtype=torch.float64, device=dev).normal_()
    dst = torch.empty_like(src)

    def copy_rate(fn_name):
        f = getattr(eng.lib, fn_name)
        call = lambda: f(n_copy, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), eng._stream())  # noqa: E731
        for _ in range(3):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            call()
        e1.record()
        e1.synchronize()
        return 2 * n_copy * 8 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9

    copy_gbs = copy_rate("fiveeq_stream_copy_f64")
    copy_wide_gbs = copy_rate("fiveeq_stream_copy_wide_f64")
    copy_nt_gbs = copy_rate("fiveeq_stream_copy_nt_f64")   # non-temporal loads and stores: the fastest plain copy of the box
    del src, dst
    roofline["stream_copy_GBs"] = copy_gbs
    roofline["stream_copy_16B_per_lane_GBs"] = copy_wide_gbs
    roofline["stream_copy_nt_GBs"] = copy_nt_gbs
    best_copy_gbs = max(copy_gbs, copy_wide_gbs, copy_nt_gbs)
    if not fusedlike:
        roofline["frac_of_stream_copy"] = roofline["achieved"] / best_copy_gbs

    # ---- the per-step kernel with NOTHING cache-resident: an ensemble whose state + parameters are several times the
    # Infinity Cache, one launch per step over all of it (chunk-major schedule off), trajectories stored -------------
    if not fusedlike and world == 1 and not a.no_hbm_resident and a.hbm_resident_members > 0:
        n_big, n_s = a.hbm_resident_members, 112
        reps = -(-n_big // n_local)
        pb = dict(p)
        for key in ("r0", "rC", "rT", "q"):
            pb[key] = p[key].repeat(1, reps)[:, :n_big].contiguous()
        big = EnsembleEngine(pb, n_big, emissions.rcp_like_emissions(n_scen, G)[250:250 + n_s], dtype=dtype, device=dev,
                             store_trajectory=not a.no_trajectory, chunk_members=0)
        w = 8 if a.dtype == "f64" else 4
        resident = w * (eng.sum_pools + 2 + 3 * G + 2) * n_big
        big.run(0, 6)
        # five batches of 100 launches, the MEDIAN batch (single passes at this size carry a hiccup of 10-40 % now and then:
        # profiles/r05/ab_variants.txt section 6)
        sm = event_timed(big, lambda t0_, t1_: big.run(t0_, t1_, join=False), 0, n_s, 100, 5,
                         lanes=big.per_step_stream_list()) / 100
        sm_med = float(np.median(sm))
        Ab = big.bytes_per_member_step("per_step")
        ach = Ab * n_big / sm_med / 1e9
        pools_c = (ctypes.c_int32 * G)(*big.pools)
        streamed = [bool(big.lib.fiveeq_rows_streamed(G, pools_c, n_, n_big, w)) for _, n_, _ in big.per_step_launches()]
        roofline["hbm_resident"] = {"members": n_big, "state_and_parameter_bytes": resident,
                                    "x_infinity_cache": resident / (256 << 20), "avg_launch_us": sm_med * 1e6,
                                    "batch_us_min_median_max": [float(sm.min()) * 1e6, sm_med * 1e6, float(sm.max()) * 1e6],
                                    "achieved": ach, "frac": ach / HBM_PEAK_GBS, "chunk_major": False,
                                    "rows": "streamed (non-temporal)" if all(streamed) else "cached",
                                    "frac_of_best_copy": ach / best_copy_gbs,
                                    "concurrent_launches": big.per_step_streams,
                                    "algorithmic_bytes_per_step": Ab * n_big}
        roofline["hbm_resident_frac"] = ach / HBM_PEAK_GBS
        big.close()
        del big, pb

    first = ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_resident_frac", "single_launch_frac",
             "single_launch_avg_us", "frac_of_stream_copy", "avg_launch_us", "kernel", "algorithmic_bytes_per_member_step",
             "members_per_launch", "concurrent_launches", "fp64_issue_frac", "fp32_issue_frac", "stream_copy_GBs",
             "stream_copy_16B_per_lane_GBs", "stream_copy_nt_GBs")
    roofline = {**{k: roofline[k] for k in first if k in roofline}, **{k: v for k, v in roofline.items() if k not in first}}
    out = {
        "metric": "ensemble_member_timesteps_per_sec", "value": value, "unit": "member-timesteps/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
        "timed_repeats": repeats, "timing": timing,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
        "data": "synthetic",
        "config": {"workload": f"{a.workload}: {desc}", "members_per_gpu": per_gpu, "members_total": n_total,
                   "gases": G, "pools": eng.pools, "scenario_steps": n_scen, "mode": a.mode, "mode_resolved": mode_run,
                   "mode_requested": mode_requested or "default",
                   "steps_per_launch": (roofline.get("steps_per_launch", 1)),
                   "trajectory_stored": eng.C is not None, "parallelism": f"member-shard x{world}",
                   "chunk_members": eng.chunk_members,
                   "collective_backend": "rccl" if backend == "nccl" else backend,
                   "control_plane": None if dist is None else "gloo over 127.0.0.1 (barriers, MAX of the clocked times)",
                   "emissions_sha256": emissions.emissions_sha256(E)[:16], "lhs_seed": params.LHS_SEED,
                   "lhs_design": "shard-computable (keyed Feistel bijection), drawn on the device",
                   "setup_s_rank0": setup_s,
                   "devices": devices},
        "roofline": roofline,
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

    summary_error = None
    if rows is not None:
        watchdog = None
        if dist is not None:
            watchdog = threading.Timer(a.summary_watchdog_s, on_timeout)
            watchdog.daemon = True
            watchdog.start()
        try:
            if os.environ.get("FIVEEQ_BENCH_HANG_SUMMARY") == str(rank):     # test hook: this rank never enters the exchange
                time.sleep(10 * a.summary_watchdog_s + 60)
            summary_stats = {}
            if dist is not None:               # data plane: RCCL ("nccl" on ROCm); its communicator comes up with the first collective
                data_group = dist.new_group(backend="nccl", timeout=tmo) if backend == "nccl" else dist.group.WORLD
            gather_summary(rows, percentiles=(5.0, 50.0, 95.0), group=data_group)       # warm: communicator + library set-up
            sync_all()
            ts = time.perf_counter()
            summ = gather_summary(rows, percentiles=(5.0, 50.0, 95.0), group=data_group, stats=summary_stats)
            torch.cuda.synchronize(dev)
            summary_ms = (time.perf_counter() - ts) * 1e3
            if watchdog is not None:
                watchdog.cancel()
            seen = {"rccl_world_size": None, "backend_seen": None}
            if dist is not None:               # as the DATA group reports them, after its first collectives have run
                seen = {"rccl_world_size": dist.get_world_size(data_group), "backend_seen": dist.get_backend(data_group)}
            if rank == 0:
                emit({"years": years, "gather_ms": summary_ms, "gather_ms_is": "second (warm) call", **seen,
                      "bytes_to_root": summary_stats.get("bytes_to_root"),
                      "bytes_to_root_per_rank": summary_stats.get("bytes_to_root_per_rank"),
                      "allreduce_bytes": summary_stats.get("allreduce_bytes"),
                      "T_mean": [float(x) for x in summ["mean"]],
                      "T_p05_p50_p95": [[float(v) for v in row] for row in summ["percentiles"]]})
        except Exception as exc:  # noqa: BLE001 - the measurement is complete: report the failure in the line, do not lose it
            summary_error = f"{type(exc).__name__}: {exc}"
            if watchdog is not None:
                watchdog.cancel()
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


if __name__ == "__main__":
    main()
