#!/usr/bin/env python3
"""bench.py — ensemble-member-timesteps/sec of the five-equation engine on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is ONE MODEL TIMESTEP of the whole ensemble = one launch of the per-timestep HIP kernel
(fiveeq_step_f64, include/fiveeq.h) on each GPU.  Workload (BASELINE.json configs[2], the
largest single-GPU configuration and the one the roofline target is quoted for): 1,000,000
members per GPU (weak scaling), CO2+CH4+N2O, fp64, deterministic RCP-like emissions
(SURVEY.md section 8d), Latin-hypercube parameter draws (seed 20261003), state and parameters
resident in HBM, C/T trajectory rows written every step.  Timesteps cycle through the 750-step
scenario (t = k mod 750); the default K + W = 750 is exactly one scenario pass.

Rank 0 prints ONE JSON line.  `value` = (members on all GPUs) x K / max-over-ranks wall time of
the K timed steps.  `roofline` prices the per-step kernel against the 8 TB/s HBM peak with the
ALGORITHMIC bytes A = w(2 SP + 4 G + 7) = 248 B per member-step; `cpu_baseline` times the CPU
oracle (plain-C port, OpenMP) on this box's host cores on a bounded sample (rank 0, N=1 only).
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
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

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
    ap.add_argument("--mode", default="per_step", choices=["per_step", "graph", "fused"])
    ap.add_argument("--no-trajectory", action="store_true", help="do not store C/T rows (drops G+1 writes from A)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-members", type=int, default=1_500_000)
    ap.add_argument("--kernel-batches", type=int, default=5, help="event-timed batches of 100 launches for roofline")
    return ap.parse_args()


def run_steps(eng, t0, k, mode):
    """Advance k model timesteps starting at scenario index t0 (cycling); returns the next index."""
    n = eng.n_steps
    t = t0 % n
    while k > 0:
        seg = min(k, n - t)
        eng.run(t, t + seg, mode=mode)
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


def cpu_baseline(kind, G, n_sample, n_steps):
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
    return {
        "value": n_sample * n_steps / dt_c, "unit": "member-timesteps/s", "cores": cores, "kind": "port",
        "sample": f"{n_sample} members x {n_steps} steps, {G} gas(es), fp64, oracle/fiveeq_oracle.c "
                  f"(gcc -O2 -fopenmp, {cores} threads of {usable} usable; final state kept, trajectories not stored), "
                  f"{dt_c:.2f} s",
        "cpu_model": model,
        "numpy_1core": {"value": n_np * n_steps / dt_np, "sample": f"{n_np} members x {n_steps} steps, "
                        f"oracle/fiveeq_oracle.py, numpy {np.__version__}, {dt_np:.2f} s"},
    }


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit(f"--gpus {a.gpus} needs the torch.distributed.run launcher (see the docstring)")
        a.gpus = world
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    # FIVEEQ_BENCH_BACKEND=gloo rehearses the multi-process path on a box with fewer GPUs than ranks
    # (ranks share devices, the summary exchange goes through host memory); the default is RCCL.
    backend = os.environ.get("FIVEEQ_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)    # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)

    from fiveeqscm_amd import emissions, params
    from fiveeqscm_amd.distributed import gather_summary, shard_bounds
    from fiveeqscm_amd.engine import EnsembleEngine

    kind, G, per_gpu, desc = WORKLOADS[a.workload]
    per_gpu = a.members or per_gpu
    n_total = per_gpu * world
    n_scen = 750
    dtype = torch.float64 if a.dtype == "f64" else torch.float32

    # global Latin hypercube over ALL members; this rank keeps its contiguous shard
    lo, hi = shard_bounds(n_total, rank, world)
    full = params.sample_ensemble(params.default_params(kind), n_total)
    p = dict(full)
    for k in ("r0", "rC", "rT", "q"):
        p[k] = np.ascontiguousarray(full[k][:, lo:hi])
    del full
    E = emissions.rcp_like_emissions(n_scen, G)
    eng = EnsembleEngine(p, hi - lo, E, dtype=dtype, device=dev, store_trajectory=not a.no_trajectory)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    # ---- device spin-up (not model work): the host spent seconds building parameters while the GPU idled
    # at its lowest clock; ~30 ms of a plain copy kernel brings it back so that a small W is enough ----
    import ctypes
    spin_src = torch.empty(1 << 25, dtype=torch.float64, device=dev).normal_()
    spin_dst = torch.empty_like(spin_src)
    for _ in range(256):
        eng.lib.fiveeq_stream_copy_f64(spin_src.numel(), ctypes.c_void_p(spin_src.data_ptr()),
                                       ctypes.c_void_p(spin_dst.data_ptr()), eng._stream())
    torch.cuda.synchronize(dev)
    del spin_src, spin_dst

    # ---- warm-up, then EXACTLY K timed steps ------------------------------------------------------
    t_idx = run_steps(eng, 0, a.warmup, a.mode)
    if a.mode == "graph":                      # instantiate the timed region's graphs outside the timing
        t_probe, k = t_idx % n_scen, a.steps
        while k > 0:
            seg = min(k, n_scen - t_probe)
            eng.prepare_graph(t_probe, t_probe + seg)
            k -= seg
            t_probe = (t_probe + seg) % n_scen
    sync_all()
    t0 = time.perf_counter()
    t_idx = run_steps(eng, t_idx, a.steps, a.mode)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    value = n_total * a.steps / elapsed

    # ---- roofline: per-launch duration of the per-step kernel, HIP events on the launch stream ----
    # The engine launches on torch's current stream, so torch.cuda.Event (hipEvent) brackets the
    # launches.  Each sample = one batch of `per_batch` launches enqueued back-to-back from C between
    # two events, divided by per_batch: the queue stays full, so the quotient is the kernel's duration
    # plus the ~1-2 us dependent-launch boundary (a single bracketed launch would add the ~10 us
    # idle-stream launch latency instead and overstate the kernel).
    A = eng.bytes_per_member_step("per_step")
    n_launch = len(eng._chunks())              # > 1 when the engine schedules chunk-major (large ensembles)
    members_per_launch = (hi - lo) / n_launch
    per_batch = 100
    samples = []
    for i in range(max(a.kernel_batches, 1)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = (t_idx + i * per_batch) % (n_scen - per_batch)
        eng.run(t, t + 5)                                   # keep the queue busy ahead of the first event
        e0.record()
        eng.run(t + 5, t + 5 + per_batch)
        e1.record()
        e1.synchronize()
        samples.append(e0.elapsed_time(e1) * 1e-3 / (per_batch * n_launch))
    samples = np.array(samples)
    k_avg = float(samples.mean())
    achieved = A * members_per_launch / k_avg / 1e9
    # achievable copy bandwidth on this box, same access shape (8 B/lane), buffers beyond the 256 MiB L3
    n_copy = 1 << 27                                        # 1 GiB read + 1 GiB written per launch
    src = torch.empty(n_copy, dtype=torch.float64, device=dev).normal_()
    dst = torch.empty_like(src)
    cp = lambda: eng.lib.fiveeq_stream_copy_f64(n_copy, ctypes.c_void_p(src.data_ptr()),   # noqa: E731
                                                ctypes.c_void_p(dst.data_ptr()), eng._stream())
    for _ in range(3):
        cp()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        cp()
    e1.record()
    e1.synchronize()
    copy_gbs = 2 * n_copy * 8 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    cpw = lambda: eng.lib.fiveeq_stream_copy_wide_f64(n_copy, ctypes.c_void_p(src.data_ptr()),   # noqa: E731
                                                      ctypes.c_void_p(dst.data_ptr()), eng._stream())
    for _ in range(3):
        cpw()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        cpw()
    e1.record()
    e1.synchronize()
    copy_wide_gbs = 2 * n_copy * 8 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            with open(tf) as fh:
                rec = json.load(fh).get(f"{a.workload}:{a.dtype}:{per_gpu}")
            traffic = rec["hbm_bytes_per_launch"] if rec else None
        except Exception:  # noqa: BLE001
            traffic = None
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": f"fiveeq::step_kernel<{'double' if a.dtype == 'f64' else 'float'},"
                          f"{','.join(str(x) for x in (eng.pools + [0, 0])[:3])}>",
                "algorithmic_bytes_per_member_step": A, "members_per_launch": members_per_launch,
                "algorithmic_bytes_per_launch": A * members_per_launch,
                "avg_launch_us": k_avg * 1e6, "min_launch_us": float(samples.min()) * 1e6,
                "launches_timed": int(samples.size) * per_batch * n_launch,
                "stream_copy_GBs": copy_gbs, "stream_copy_16B_per_lane_GBs": copy_wide_gbs,
                "frac_of_stream_copy": achieved / max(copy_gbs, copy_wide_gbs)}

    # ---- end-of-run exchange (the only collective): summary statistics of T over all members ------
    summary, summary_error = None, None
    if eng.T is not None:
        torch.cuda.synchronize(dev)
        ts = time.perf_counter()
        done = min(a.warmup + a.steps, n_scen)                  # scenario steps the timed run has written
        years = [t for t in (249, 499, 749) if t < done] or [done - 1]
        try:
            summary = gather_summary(eng.T[years], percentiles=(5.0, 50.0, 95.0))
        except Exception as exc:  # noqa: BLE001 - the exchange is outside the timed region: report, do not lose the line
            summary = None
            summary_error = f"{type(exc).__name__}: {exc}"
        torch.cuda.synchronize(dev)
        summary_ms = (time.perf_counter() - ts) * 1e3

    out = {
        "metric": "ensemble_member_timesteps_per_sec", "value": value, "unit": "member-timesteps/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
        "data": "synthetic",
        "config": {"workload": f"{a.workload}: {desc}", "members_per_gpu": per_gpu, "members_total": n_total,
                   "gases": G, "pools": eng.pools, "scenario_steps": n_scen, "mode": a.mode,
                   "trajectory_stored": eng.C is not None, "parallelism": f"member-shard x{world}",
                   "chunk_members": eng.chunk_members,
                   "collective_backend": "rccl" if backend == "nccl" else backend,
                   "emissions_sha256": emissions.emissions_sha256(E)[:16], "lhs_seed": params.LHS_SEED},
        "roofline": roofline,
    }
    if summary is not None and rank == 0:
        out["summary"] = {"years": years, "gather_ms": summary_ms,
                          "T_mean": [float(x) for x in summary["mean"]],
                          "T_p05_p50_p95": [[float(v) for v in row] for row in summary["percentiles"]]}
    if summary_error is not None and rank == 0:
        out["summary"] = {"error": summary_error}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(kind, G, a.cpu_sample_members, n_scen)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
