"""bench.py's contract (one JSON line with the keys the driver and the judge read), exercised on small arguments."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timed_s="0.05", live_traffic=False):
    """One plain `python bench.py ...` call.  --timed-s: these tests check the line, not the figure, so they clock 50 ms of
    device time instead of the default 6.5 s (the driver-call test below keeps the default); the two rocprofv3 --pmc child passes
    that measure roofline.traffic (25 s) run in the contract test only."""
    extra = [] if timed_s is None or "--timed-s" in args else ["--timed-s", timed_s]
    extra += [] if live_traffic or "--no-live-traffic" in args else ["--no-live-traffic"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, *extra], capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_default_mode_line_has_the_contract_keys():
    d = _bench("--steps", "60", "--warmup", "5", "--members", "300000", "--cpu-sample-members", "20000",
               "--hbm-resident-members", "2000000", "--kernel-batches", "1", live_traffic=True)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "summary"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 60 and d["warmup"] == 5 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert abs(d["value"] - 300000 * 60 / (d["ms_per_step"] * 60e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.2
    assert abs(r["algorithmic_bytes_per_member_step"] - 248.0) < 1e-9 and r["kernel"] == "fiveeq::step_kernel<double,4,1,1>"
    assert r["avg_launch_us"] * 1e-3 <= d["ms_per_step"] * 1.25                 # kernel time consistent with the wall figure
    assert 0.0 < r["hbm_resident_frac"] < 1.0 and r["hbm_resident"]["x_infinity_cache"] > 1.0
    # the beyond-the-cache rate depends on where the driver puts the stored-trajectory buffers: several placements, the median
    pl = sorted(r["hbm_resident"]["placements"]["frac_each"])
    assert len(pl) == 5 and pl[0] > 0.3 and r["hbm_resident_frac"] == pl[2] == r["hbm_resident"]["frac"]
    # north_star's literal shape (one launch per timestep on one stream) beside the default; 300k members run as one launch
    # anyway, so the two figures are one measurement here
    assert r["single_launch"]["frac"] == r["single_launch_frac"] and 0.0 < r["single_launch_frac"] < 1.2
    assert (r["concurrent_launches"] == 1) == (r["single_launch_avg_us"] == r["avg_launch_us"])
    # the scalars a reader needs come FIRST in the object (the driver's parser keeps the head of a nested object)
    assert list(r)[:9] == ["bound", "achieved", "peak", "unit", "frac", "regime", "traffic", "hbm_resident_frac", "single_launch_frac"]
    # `frac` prices algorithmic bytes against the HBM PEAK: the regime beside it says whether those bytes cross HBM at all
    # (300k members: 46 MB of rows live in the Infinity Cache; the beyond-the-cache leg carries its own label)
    assert r["regime"].startswith("infinity-cache-resident (46 MB") and r["hbm_resident"]["regime"].startswith("hbm-streamed (")
    assert d["config"]["mode_resolved"] == "per_step" and len(d["config"]["devices"]) == 1
    dev0 = d["config"]["devices"][0]
    assert dev0["rank"] == 0 and dev0["device_index"] == 0 and dev0["name"] and len(dev0["pci_bus_id"].split(":")) == 3
    # the rank's host thread was bound to its GPU's CPUs from sysfs before the GPU was touched, and the runtime's own PCI
    # address confirmed the guess (or the report says why nothing was applied: a sandbox without the KFD topology)
    cb = dev0["cpu_binding"]
    assert cb["applied"] is True or cb.get("reason"), cb
    if cb["applied"]:
        assert dev0["cpus"] == cb["cpus"] and cb["n_cpus"] >= 1 and cb["pci_bus_id_runtime"] == dev0["pci_bus_id"]
        assert cb["verified"] or cb.get("rebound_after_init")
    ss = dev0["side_streams"]                                    # 300k members: one launch per step, no side stream wanted
    assert ss["wanted"] == 0 and ss["probe_enabled"] is True
    assert len(d["timing"]["per_rank_ms_per_step"]) == 1 and abs(d["timing"]["per_rank_ms_per_step"][0] - d["ms_per_step"]) < 1e-9
    assert "workload" in d["config"] and "model" not in d["config"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert c["numpy_nproc"]["processes"] >= 1 and c["c_port_openmp"]["value"] > 0 and c["numpy_1core"]["value"] > 0
    assert c["value"] == max(c["numpy_nproc"]["value"], c["c_port_openmp"]["value"])          # the stronger whole-box figure
    # roofline.traffic is MEASURED BY THIS RUN: two child rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same kernel at the
    # same size before the GPU was touched, calibrated on a known copy — within a few per cent of the algorithmic bytes per launch
    assert r["traffic_source"].startswith("measured by this run: rocprofv3 --pmc FETCH_SIZE"), r.get("traffic_live_error")
    assert 0.95 * r["algorithmic_bytes_per_launch"] <= r["traffic"] <= 1.15 * r["algorithmic_bytes_per_launch"]
    td = r["traffic_detail"]
    assert td["step_dispatches"][0] >= 60 and td["copy_calibration"]["copy_dispatches"][0] >= 5 and td["seconds"] < 120
    assert d["timed_repeats"] == 1 or d["timing"]["wall_ms_per_step_over_all_repeats"] >= d["ms_per_step"] * 0.98
    t = d["timing"]
    assert t["device_s_clocked"] >= 0.05 and 0.0 < t["host_enqueue_us_per_step_min"] <= t["host_enqueue_us_per_step"]
    assert abs(t["host_share"] - t["host_enqueue_us_per_step"] * 1e-3 / d["ms_per_step"]) < 1e-9 and t["host_share"] < 1.0
    s = d["summary"]
    assert len(s["T_mean"]) == len(s["years"]) and s["bytes_to_root"] == 0 and s["gather_ms"] > 0


def test_fused_family_lines_price_their_own_kernel():
    d = _bench("--steps", "740", "--warmup", "10", "--members", "200000", "--mode", "fused", "--no-cpu-baseline",
               "--kernel-batches", "2")
    r = d["roofline"]
    assert r["bound"] == "fp64-valu" and r["kernel"].startswith("fiveeq::fused_kernel<double") and "cpu_baseline" not in d
    assert r["algorithmic_bytes_per_member_step"] < 40.0 and r["steps_per_launch"] == 128      # 200k members: relaunched (fused_span)
    # BASELINE configs[1]: --mode auto is the small-ensemble kernel, one member per quad of lanes
    d2 = _bench("--workload", "config2", "--mode", "auto", "--no-cpu-baseline", "--kernel-batches", "2")
    assert d2["config"]["mode"] == "auto" and d2["config"]["mode_resolved"] == "small" and d2["config"]["steps_per_launch"] == 750
    r2 = d2["roofline"]
    assert r2["kernel"] == "fiveeq::small_kernel<double,4,4,false>" and r2["lanes_per_member"] == 4 and r2["waves"] == 625
    d3 = _bench("--workload", "config2", "--no-cpu-baseline", "--kernel-batches", "1", "--no-hbm-resident")
    assert d2["value"] > 3.0 * d3["value"]                      # ... against the launch-bound per-step form
    d4 = _bench("--workload", "config2", "--mode", "ksteps", "--no-cpu-baseline", "--kernel-batches", "2")
    assert d4["roofline"]["kernel"].startswith("fiveeq::fused_kernel<double,4,0,0") and d2["value"] > 1.3 * d4["value"]


# ---- the N > 1 path: exactly what the driver launches on a multi-GPU node, rehearsed on the one GPU of the test box ----
_SMALL = ("--steps", "20", "--warmup", "5", "--members", "200000", "--no-cpu-baseline", "--no-hbm-resident",
          "--kernel-batches", "1", "--timed-s", "0.05", "--no-live-traffic")
_RANK_ENV = dict(FIVEEQ_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")


def _torchrun(n, *args, env=None, timeout=900):
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py --gpus n ...` as a fresh child
    process; the ranks share the card and exchange over gloo (FIVEEQ_BENCH_BACKEND), the launch line is the driver's."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), *args]
    full_env = dict(os.environ, **_RANK_ENV)
    full_env.update(env or {})
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=full_env)


def _plain(n, *args, env=None, timeout=900):
    """`python3 bench.py --gpus n ...` with NO launcher around it — the form the driver uses for --gpus 1.  For n > 1 bench.py
    starts the ranks itself (a child torch.distributed.run, before it touches any GPU)."""
    full_env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    full_env.update(_RANK_ENV)
    full_env.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), *args], capture_output=True,
                          text=True, timeout=timeout, env=full_env)


def _summaries_agree(a_, b_, rel=1e-12):
    flat = lambda v: [y for x in v for y in (x if isinstance(x, list) else [x])]     # noqa: E731
    for key in ("T_mean", "T_p05_p50_p95"):
        got, want = flat(a_[key]), flat(b_[key])
        assert len(got) == len(want) and all(abs(g - w) <= rel * abs(w) for g, w in zip(got, want)), (key, a_[key], b_[key])


def _one_line(out):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 4])
def test_multi_rank_launch_prints_one_line_and_the_world_size_invariant_summary(n):
    d = _one_line(_torchrun(n, *_SMALL))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "summary", "timed_repeats"):
        assert key in d, key
    assert d["n_gpus"] == n and d["steps"] == 20 and d["warmup"] == 5 and "cpu_baseline" not in d
    assert d["config"]["members_total"] == n * 200000 and d["config"]["collective_backend"] == "gloo"
    assert d["timed_repeats"] >= 3 and d["timed_repeats"] % 2 == 1          # a 20-step block is well under 50 ms
    assert d["config"]["control_plane"].startswith("gloo") and 0.0 < d["timing"]["host_share"]
    assert abs(d["value"] - n * 200000 * 20 / (d["ms_per_step"] * 20e-3)) < 1e-6 * d["value"]
    lo_ms, med_ms, hi_ms = d["timing"]["block_ms_min_median_max"]
    assert lo_ms <= med_ms <= hi_ms and abs(med_ms - d["ms_per_step"] * 20) < 1e-9 * med_ms
    assert d["summary"]["bytes_to_root"] > 0 and d["summary"]["years"] == [249, 499, 749]
    # the line proves what ran: n distinct ranks, their devices, their own timings, and the exchange group as it reports itself
    devs = d["config"]["devices"]
    assert [x["rank"] for x in devs] == list(range(n)) and len({x["pid"] for x in devs}) == n
    assert all(x["name"] and x["device_index"] == x["local_rank"] % x["visible_devices"] for x in devs)
    # every rank's host thread on its own CPUs: the ranks share the one card here, so they share its CPU list in disjoint slices
    bound = [x["cpu_binding"] for x in devs if x["cpu_binding"]["applied"]]
    if bound:
        from fiveeqscm_amd.hostbind import parse_cpulist
        sets = [set(parse_cpulist(x["cpus"])) for x in bound]
        assert len(bound) == n and all(sets) and all(not (sets[i] & sets[j]) for i in range(n) for j in range(i)), [x["cpus"] for x in bound]
    t = d["timing"]
    assert len(t["per_rank_ms_per_step"]) == n == len(t["per_rank_host_enqueue_us"]) and min(t["per_rank_ms_per_step"]) > 0
    assert max(t["per_rank_ms_per_step"]) <= d["ms_per_step"] * 1.5 and max(t["per_rank_host_enqueue_us"]) <= t["host_enqueue_us_per_step"] * 1.0001
    sm = d["summary"]
    assert sm["rccl_world_size"] == n and sm["backend_seen"] == "gloo"             # (the rehearsal's data group is gloo)
    assert len(sm["bytes_to_root_per_rank"]) == n and sm["bytes_to_root_per_rank"][0] == 0
    assert sum(sm["bytes_to_root_per_rank"]) == sm["bytes_to_root"] and all(b > 0 for b in sm["bytes_to_root_per_rank"][1:])
    # ONE design for every world size (the shard-computable Latin hypercube) and members never interact: the summary
    # of n ranks x 200k members is the summary of one rank with n x 200k members.  End to end through the launcher,
    # the sharding, the kernels and the exchange.
    one = _bench("--steps", "20", "--warmup", "5", "--members", str(n * 200000), "--no-cpu-baseline", "--no-hbm-resident",
                 "--kernel-batches", "1")
    assert one["n_gpus"] == 1 and one["summary"]["years"] == [249, 499, 749]
    _summaries_agree(d["summary"], one["summary"])
    # ... and the PLAIN form, `python3 bench.py --gpus n` with no launcher (how the driver calls --gpus 1): bench.py starts
    # the ranks itself and the one line that comes back carries the same summary
    pl = _one_line(_plain(n, *_SMALL))
    assert pl["n_gpus"] == n and pl["config"]["members_total"] == n * 200000 and pl["config"]["parallelism"] == f"member-shard x{n}"
    _summaries_agree(pl["summary"], d["summary"])


def test_one_rank_through_the_launcher_is_the_plain_call():
    d = _one_line(_torchrun(1, *_SMALL))
    one = _bench(*_SMALL)
    assert d["n_gpus"] == one["n_gpus"] == 1 and d["config"]["control_plane"] is None
    _summaries_agree(d["summary"], one["summary"], rel=0.0)


@pytest.mark.parametrize("launch", ["launcher", "plain"])
def test_a_dead_rank_fails_the_launch(launch):
    run = _torchrun if launch == "launcher" else _plain
    out = run(2, *_SMALL, env={"FIVEEQ_BENCH_FAIL_RANK": "1"}, timeout=600)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_a_hung_summary_exchange_cannot_lose_the_measured_line():
    """Rank 1 never enters the end-of-run exchange (test hook), so rank 0 sits in a collective that cannot complete —
    what a first RCCL contact across xGMI gone wrong looks like.  The watchdog prints the complete line with summary.error
    after --summary-watchdog-s and the job exits non-zero, long before any launcher or driver limit."""
    import time
    t0 = time.time()
    out = _plain(2, *_SMALL, "--summary-watchdog-s", "8", env={"FIVEEQ_BENCH_HANG_SUMMARY": "1"}, timeout=300)
    assert out.returncode != 0 and time.time() - t0 < 200
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads(lines[0])
    assert "timeout" in d["summary"]["error"] and d["value"] > 0 and d["n_gpus"] == 2 and d["roofline"]["frac"] > 0


def test_an_rccl_failure_on_first_contact_is_reported_and_the_exchange_repeated_over_gloo():
    """Two ranks over RCCL on the ONE GPU of the test box: RCCL refuses two ranks on one device when its communicator comes up
    — a real first-contact failure, on every rank alike.  That happens in the summary section, after the measurement: every rank
    reports it over the control plane, ALL repeat the exchange over gloo, and rank 0 prints the complete line — the summary of the
    whole ensemble, the RCCL error beside it — and the job exits 0 (round 6: a multi-GPU measurement is not held hostage by the
    one collective that is not part of it; a HUNG exchange still ends in the watchdog, the test above)."""
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs a box where two ranks must share one GPU")
    out = _plain(2, *_SMALL, "--summary-watchdog-s", "60", env={"FIVEEQ_BENCH_BACKEND": "nccl"}, timeout=400)
    d = _one_line(out)
    sm = d["summary"]
    assert "error" not in sm and sm["rccl_error"] and len(sm["rccl_errors_per_rank"]) == 2 and all(sm["rccl_errors_per_rank"])
    assert sm["backend_seen"] == "gloo" and sm["rccl_world_size"] == 2 and "gloo" in sm["fallback"]
    assert d["value"] > 0 and d["n_gpus"] == 2 and d["config"]["collective_backend"] == "rccl"
    assert d["config"]["control_plane"].startswith("gloo") and d["timing"]["host_share"] > 0
    ref = _one_line(_plain(2, *_SMALL))                                      # the same job with gloo as its data plane from the start
    _summaries_agree(sm, ref["summary"], rel=0.0)
    print("RCCL first-contact failure reported as:", sm["rccl_error"][:200])


def test_a_host_bound_multi_rank_run_falls_back_to_graph_replay():
    """N > 1, no explicit --mode: the line reports whether the slowest rank's host thread needs more than --host-share-limit of a
    step to enqueue it (`would_switch`); the default keeps the per-step form at every N (the scaling curve compares one launch
    form with itself), and only --host-fallback switches the timed region to the hipGraph replay of the same launches (same
    kernels, same bits).  Forced here with a limit of 0 and excluded with a limit of 100; an explicit --mode is never overridden."""
    reported = _one_line(_plain(2, *_SMALL, "--host-share-limit", "0"))
    fb = reported["timing"]["host_fallback"]
    assert reported["config"]["mode"] == "per_step" and fb["would_switch"] is True and fb["enabled"] is False and fb["switched_to_graph"] is False
    forced = _one_line(_plain(2, *_SMALL, "--host-share-limit", "0", "--host-fallback"))
    assert forced["config"]["mode"] == "graph" and forced["config"]["mode_requested"] == "default"
    fb = forced["timing"]["host_fallback"]
    assert fb["switched_to_graph"] is True and fb["per_step_host_share"] > 0 and fb["limit"] == 0.0 and fb["enabled"] is True
    kept = _one_line(_plain(2, *_SMALL, "--host-share-limit", "100", "--host-fallback"))     # a limit no box reaches: never switches
    assert kept["config"]["mode"] == "per_step" and kept["timing"]["host_fallback"]["switched_to_graph"] is False
    default = _one_line(_plain(2, *_SMALL))                                          # the default: reported, not acted on
    fb_d = default["timing"]["host_fallback"]
    assert fb_d["limit"] == 0.5 and fb_d["would_switch"] == (fb_d["per_step_host_share"] >= 0.5) and default["config"]["mode"] == "per_step"
    _summaries_agree(forced["summary"], kept["summary"], rel=0.0)                  # bit-identical kernels
    assert forced["timing"]["host_enqueue_us_per_step"] < kept["timing"]["host_enqueue_us_per_step"]
    explicit = _one_line(_plain(2, *_SMALL, "--mode", "per_step", "--host-share-limit", "0", "--host-fallback"))
    assert explicit["config"]["mode"] == "per_step" and explicit["timing"]["host_fallback"] is None
    one = _bench(*_SMALL)
    assert one["config"]["mode"] == "per_step" and one["config"]["mode_requested"] == "default" and one["timing"]["host_fallback"] is None


def test_four_ranks_enqueue_at_once_and_the_host_keeps_up():
    """The host side of north_star's >= 7x at 8 GPUs: every rank is one Python thread issuing 2 launches per 35 us (3.5 us of
    host time per hipLaunchKernel, profiles/r04/host_enqueue_profile.txt).  Four ranks of the driver's own workload — 1M
    members each; with the launcher's agent and this test process that is the six processes the GPU pool allows on one card
    (tools/rehearse_multi_gpu.sh goes to five outside pytest) — enqueue their bursts
    behind a common barrier; the slowest rank's enqueue time per step must stay under half a step of ONE un-shared GPU (the
    ranks share the card here, so the line's own step time is ~4x longer and its `host_share` would flatter the ratio)."""
    args = ("--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-hbm-resident", "--kernel-batches", "1", "--timed-s", "0.2")
    out = _plain(4, *args)
    d = _one_line(out)
    assert [ln for ln in out.stdout.splitlines() if ln.strip()] == [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    solo = _bench(*args)
    t = d["timing"]
    assert d["n_gpus"] == 4 and d["config"]["members_per_gpu"] == 1_000_000 and t["host_enqueue_us_per_step"] > 0
    share = t["host_enqueue_us_per_step"] * 1e-3 / solo["ms_per_step"]
    print(f"4 ranks x 1M members: host enqueue {t['host_enqueue_us_per_step']:.2f} us/step (min "
          f"{t['host_enqueue_us_per_step_min']:.2f}; one rank alone {solo['timing']['host_enqueue_us_per_step']:.2f}); one "
          f"un-shared 1M-member step {solo['ms_per_step'] * 1e3:.2f} us -> host share with 4 ranks enqueuing at once {share:.3f}")
    assert share < 0.5
    one = _bench("--steps", "20", "--warmup", "5", "--members", "4000000", "--no-cpu-baseline", "--no-hbm-resident",
                 "--kernel-batches", "1")
    _summaries_agree(d["summary"], one["summary"])


def test_one_rank_job_runs_every_collective_over_rccl():
    """FIVEEQ_BENCH_FORCE_DIST=1: bench.py builds a one-rank RCCL ("nccl") group and runs the barriers, the MAX of the
    block times and the summary exchange on it — the N > 1 code of this file on the device backend, on one GPU."""
    env = dict(os.environ, FIVEEQ_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(s.getsockname()[1])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *_SMALL], capture_output=True, text=True,
                         timeout=600, env=env)
    d = _one_line(out)
    plain = _bench(*_SMALL)
    assert d["n_gpus"] == 1 and d["config"]["collective_backend"] == "rccl" and d["summary"]["allreduce_bytes"] > 0
    assert d["summary"]["rccl_world_size"] == 1 and d["summary"]["backend_seen"] == "nccl"        # as the RCCL group reports itself
    assert plain["summary"]["allreduce_bytes"] == 0 and plain["summary"]["rccl_world_size"] is None
    assert d["summary"]["T_mean"] == pytest.approx(plain["summary"]["T_mean"], rel=1e-12)
    for a_, b_ in zip(d["summary"]["T_p05_p50_p95"], plain["summary"]["T_p05_p50_p95"]):
        assert a_ == pytest.approx(b_, rel=1e-12)


def test_a_twenty_step_call_reports_what_a_full_pass_reports():
    """The driver's call is --steps 20 --warmup 5: the median of the repeated 20-step blocks must be the throughput of
    a whole 740-step pass (within box noise; the review's mark is 2 %, the assertion allows 5 %)."""
    short = _bench("--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-hbm-resident", "--kernel-batches", "1",
                   timed_s=None)                                # the driver's call: the default 6.5 s of clocked device time
    full = _bench("--steps", "740", "--warmup", "10", "--no-cpu-baseline", "--no-hbm-resident", "--kernel-batches", "1",
                  "--timed-s", "0")                             # ONE 740-step block on the wall clock
    assert short["timed_repeats"] > 1000 and short["timing"]["device_s_clocked"] > 6.0 and full["timed_repeats"] == 1
    print(f"20-step median {short['value']:.4g} vs 740-step {full['value']:.4g}: ratio {short['value'] / full['value']:.4f}")
    assert abs(short["value"] / full["value"] - 1.0) < 0.05
