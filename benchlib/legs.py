"""Measurement legs of bench.py: everything the line reports BESIDE `value` — the roofline of the timed mode's kernel (HIP events
on the launch streams), north_star's literal one-launch shape, the box's plain copy rates, the per-step kernel on an ensemble far
beyond the Infinity Cache (`hbm_resident`), the VALU issue fraction, and the host side of a step.  None of this is inside the
clocked region; bench.py calls these after `value` is fixed."""
import ctypes
import json
import os
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
INFINITY_CACHE = 256 << 20
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
N_SCEN = 750

ROOFLINE_FIRST = ("bound", "achieved", "peak", "unit", "frac", "regime", "traffic", "hbm_resident_frac", "single_launch_frac",
                  "single_launch_avg_us", "frac_of_stream_copy", "avg_launch_us", "kernel", "algorithmic_bytes_per_member_step",
                  "members_per_launch", "concurrent_launches", "fp64_issue_frac", "fp32_issue_frac", "stream_copy_GBs",
                  "stream_copy_16B_per_lane_GBs", "stream_copy_nt_GBs")


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


def prepare_graphs(eng, t_from, k):
    """Instantiate (outside any timing) the hipGraph plans of k steps from scenario index t_from."""
    while k > 0:
        seg = min(k, eng.n_steps - t_from)
        eng.prepare_graph(t_from, t_from + seg)
        k -= seg
        t_from = (t_from + seg) % eng.n_steps


def spin_up(eng, dev):
    """Device spin-up (not model work): the GPU idles at its lowest clock during host set-up; ~30 ms of a plain copy kernel brings
    it back so that a small W is enough."""
    src = torch.empty(1 << 25, dtype=torch.float64, device=dev).normal_()
    dst = torch.empty_like(src)
    for _ in range(256):
        eng.lib.fiveeq_stream_copy_f64(src.numel(), ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), eng._stream())
    torch.cuda.synchronize(dev)


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


def host_enqueue(eng, ctl, mode, t_from, k_burst, k_steps):
    """The HOST side of a step: how long this rank's CPU thread needs to ENQUEUE one timestep (Python + ctypes + the
    hipLaunchKernel calls inside fiveeq_run_*), measured on a drained device with a short burst so that the HIP queue never fills
    (a full queue blocks the caller: that would clock the device, not the host).  Every sample starts behind a barrier, so with N
    ranks all N host threads enqueue AT THE SAME TIME — the contention an 8-GPU node's host side sees.
    Returns (median, min) seconds per step over 15 bursts, MAX over ranks; the device time per step of the same bursts (HIP events,
    MAX over ranks, median); the next scenario index; this rank's own median."""
    enq, dev_t = [], []
    for _ in range(15):
        if mode == "graph":
            prepare_graphs(eng, t_from % eng.n_steps, k_burst)
        ctl.sync_all()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        t_from = run_steps(eng, t_from, k_burst, mode, k_steps, join=False)
        enq.append((time.perf_counter() - t0) / k_burst)
        eng.join()
        e1.record()
        e1.synchronize()
        dev_t.append(e0.elapsed_time(e1) * 1e-3 / k_burst)
    torch.cuda.synchronize(ctl.dev)
    mine = float(np.median(enq))
    med, mn, dmed = ctl.max_over_ranks([mine, float(np.min(enq)), float(np.median(dev_t))])
    return med, mn, dmed, t_from, mine


def load_profile_json(name, key):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None
    try:
        with open(path) as fh:
            return json.load(fh).get(key)
    except Exception:  # noqa: BLE001
        return None


def kernel_tags(eng, a, n_local):
    """Names the roofline legs share: (template type name, lane type name, valu.json tag, members per wave, 'P0,P1,P2', packed)."""
    tname = "double" if a.dtype == "f64" else "float"
    # fp32 runs the packed kernels (two members per lane) whenever the rows allow 8-byte accesses: even members per launch
    packed = a.dtype == "f32" and n_local % 2 == 0 and (eng.chunk_members % 2 == 0)
    lname = "float2 (two members per lane)" if packed else tname
    pools3 = ",".join(str(x) for x in (eng.pools + [0, 0])[:3])
    return tname, lname, a.dtype + ("x2" if packed else ""), 128 if packed else 64, pools3, packed


def regime(resident_bytes, n_seq=1):
    """Where the rows of one launch live between two launches: the label that goes beside `frac` (a fraction of the HBM PEAK is
    an HBM fraction only when the bytes cross HBM)."""
    mb, share = resident_bytes / 1e6, resident_bytes / INFINITY_CACHE
    if n_seq > 1:
        return (f"chunk-major ({mb:.0f} MB per chunk, {share:.2f} of the Infinity Cache: each chunk's rows stay cached between its launches, "
                f"{n_seq} chunks one after the other)")
    if share <= 0.8:
        return f"infinity-cache-resident ({mb:.0f} MB of state + parameters in the 256 MiB Infinity Cache; hbm_resident_frac is the HBM-true figure)"
    return f"hbm-streamed ({mb:.0f} MB of state + parameters per launch, {share:.1f}x the 256 MiB Infinity Cache)"


def per_step_roofline(eng, a, G, per_gpu, n_local, t_idx):
    """Per-launch duration of the per-step kernel, HIP events on the launch streams.  The engine launches on torch's current
    stream, so torch.cuda.Event (hipEvent) brackets the launches.  Each sample = one batch of launches enqueued back-to-back from
    C between two events: the queue stays full, so the quotient is the kernel's duration plus the ~1-2 us dependent-launch
    boundary (a single bracketed launch would add the ~10 us idle-stream launch latency instead and overstate the kernel).
    Returns (roofline dict, avg seconds per launch, valu.json key)."""
    _, lname, vtag, _, pools3, _ = kernel_tags(eng, a, n_local)
    wbytes = 8 if a.dtype == "f64" else 4
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
    samples = event_timed(eng, lambda t0_, t1_: eng.run(t0_, t1_, mode="per_step", join=False), t_idx, N_SCEN, per_batch,
                          a.kernel_batches, lanes=eng.per_step_stream_list())
    samples = samples / (per_batch * n_seq)
    k_avg = float(samples.mean())
    achieved = A * members_per_launch * conc / k_avg / 1e9
    tkey = f"{a.workload}:{a.dtype}:{per_gpu}"
    traffic = (load_profile_json("traffic.json", tkey) or {}).get("hbm_bytes_per_launch")
    resident = wbytes * (eng.sum_pools + 2 + 3 * G + 2) * members_per_launch * conc   # state + parameter rows of a chunk
    if resident <= 0.8 * INFINITY_CACHE:
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
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "regime": regime(resident, n_seq),
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
        one = event_timed(eng, lambda t0_, t1_: eng.run(t0_, t1_, mode="per_step", join=False), t_idx, N_SCEN, per_batch,
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
    return roofline, k_avg, f"step:{vtag}:{pools3}"


def fused_roofline(eng, a, mode_run, k_steps, n_local):
    """The time-fused family: one launch covers `span` steps; priced per step with its own A.  Timed the way the timed region
    runs it: whole scenario passes from the initial state (HIP events on the launch stream around each pass; the launches of a
    pass are enqueued back-to-back from C).  Returns (roofline dict, avg seconds per step in the kernel, valu.json key,
    members per wave)."""
    tname, lname, vtag, members_per_wave, pools3, _ = kernel_tags(eng, a, n_local)
    valu_peak = SIMDS * CLOCK_HZ / VALU_CYCLES_PER_INSTR[a.dtype]
    lpm, single = None, False
    if mode_run == "small":
        lpm = eng.small_form()
        span, kname, mode_t = N_SCEN, "small_kernel", "small"
        single = len(eng.pools) == 1
        lname = f"{tname},{eng.pools[0]},{lpm}" if single else f"{tname},{pools3}"
        vtag, members_per_wave = a.dtype, 64 // lpm                  # (never packed)
    elif mode_run == "fused":
        span, kname, mode_t = eng.fused_span_steps(N_SCEN), "fused_kernel", "fused"     # the engine relaunches small ensembles
    else:
        span = k_steps or eng.auto_k_steps()
        kname, mode_t = "fused_kernel", "ksteps"
    A = eng.bytes_per_member_step(mode_t, span if mode_t == "ksteps" else None)
    samples = []
    for _ in range(max(a.kernel_batches, 2)):
        eng.reset_state()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.run(0, N_SCEN, mode=mode_t, k_steps=span if mode_t == "ksteps" else None)
        e1.record()
        e1.synchronize()
        samples.append(e0.elapsed_time(e1) * 1e-3 / N_SCEN)
    samples = np.array(samples[1:])                         # the first pass re-warms
    reps = -(-N_SCEN // span)
    k_avg = float(samples.mean())                           # seconds per model step inside the kernel
    achieved = A * n_local / k_avg / 1e9
    kernel_name = ((f"fiveeq::small_kernel<{lname},false>" if single else
                    (f"fiveeq::small_octet_kernel<{tname}>" if lpm == 8 else f"fiveeq::small_multi_kernel<{lname},false>"))
                   if kname == "small_kernel" else f"fiveeq::{kname}<{lname},{pools3}>")
    roofline = {"bound": "fp64-valu" if a.dtype == "f64" else "fp32-valu", "unit": "wave-instr/s",
                "achieved": None, "peak": valu_peak, "frac": None, "traffic": None,
                "regime": "register-resident state (time-fused): bound by VALU issue, not by memory",
                "kernel": kernel_name, "steps_per_launch": span,
                "algorithmic_bytes_per_member_step": A, "members_per_launch": n_local,
                "hbm_GBs_of_algorithmic_bytes": achieved, "hbm_frac": achieved / HBM_PEAK_GBS,
                "avg_step_us_in_kernel": k_avg * 1e6, "launches_timed": int(samples.size) * reps,
                "timed_as": f"{samples.size} whole {N_SCEN}-step scenario passes from the initial state",
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
    if eng.compensated:                                     # its own instruction stream: no committed SQ-counter pass under this key
        roofline["kernel"], kkey = kernel_name.replace(">", ",false,false,true> (compensated fp32)"), kkey + ":comp"
    return roofline, k_avg, kkey, members_per_wave


def add_valu_issue(roofline, a, kkey, k_avg, members_per_wave, packed, fusedlike):
    """VALU issue: instructions per wave-step from the committed SQ-counter pass (profiles/valu.json, produced by
    tools/collect_profiles.sh with rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES ...), times the waves this bench ran."""
    valu = load_profile_json("valu.json", kkey)
    if not valu:
        return
    valu_peak = SIMDS * CLOCK_HZ / VALU_CYCLES_PER_INSTR[a.dtype]
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


def copy_rates(eng, dev, roofline, fusedlike):
    """Achievable copy bandwidth on this box, same access shape (8 B/lane), buffers beyond the 256 MiB L3.  Returns the best."""
    n_copy = 1 << 27                                        # 1 GiB read + 1 GiB written per launch
    src = torch.empty(n_copy, dtype=torch.float64, device=dev).normal_()
    dst = torch.empty_like(src)

    def rate(fn_name):
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

    roofline["stream_copy_GBs"] = rate("fiveeq_stream_copy_f64")
    roofline["stream_copy_16B_per_lane_GBs"] = rate("fiveeq_stream_copy_wide_f64")
    roofline["stream_copy_nt_GBs"] = rate("fiveeq_stream_copy_nt_f64")   # non-temporal loads and stores: the fastest plain copy of the box
    best = max(roofline["stream_copy_GBs"], roofline["stream_copy_16B_per_lane_GBs"], roofline["stream_copy_nt_GBs"])
    if not fusedlike:
        roofline["frac_of_stream_copy"] = roofline["achieved"] / best
    return best


def hbm_resident(eng, a, p, G, dtype, dev, n_local, best_copy_gbs, roofline):
    """The per-step kernel with NOTHING cache-resident: an ensemble whose state + parameters are several times the Infinity
    Cache, one launch per step over all of it (chunk-major schedule off), trajectories stored."""
    from fiveeqscm_amd import emissions
    from fiveeqscm_amd.engine import EnsembleEngine
    n_big, n_s = a.hbm_resident_members, 112
    reps = -(-n_big // n_local)
    pb = dict(p)
    for key in ("r0", "rC", "rT", "q"):
        pb[key] = p[key].repeat(1, reps)[:, :n_big].contiguous()
    big = EnsembleEngine(pb, n_big, emissions.rcp_like_emissions(N_SCEN, G)[250:250 + n_s], dtype=dtype, device=dev,
                         store_trajectory=not a.no_trajectory, chunk_members=0)
    w = 8 if a.dtype == "f64" else 4
    resident = w * (eng.sum_pools + 2 + 3 * G + 2) * n_big
    # `--hbm-placements` PLACEMENTS of the stored-trajectory buffers (round 6, profiles/r06/hbm_low_mode.txt): this rate moves
    # between 0.69 and 0.80 of 8 TB/s with WHERE the driver puts C and T — the write-only rows; re-allocating them inside one
    # process flips it, re-placing r / q / R / S does not, no L2 / EA / TLB counter or clock tells the placements apart — so one
    # allocation is one draw.  The leg measures the engine as constructed, then re-allocates C and T (behind a random spacer, the
    # old buffers returned to the driver) and measures again; `frac` is the MEDIAN placement, all of them are listed.
    rng = np.random.default_rng(os.getpid())
    Ab = big.bytes_per_member_step("per_step")
    per_placement, spacers = [], []
    for k in range(max(1, a.hbm_placements) if big.T is not None else 1):      # (no stored rows: nothing to re-place)
        if k > 0:
            spacers = spacers[-2:] + [torch.empty(int(rng.integers(1, 2048)) << 20, dtype=torch.uint8, device=dev)]
            for name in ("C", "T"):
                old = getattr(big, name)
                if old is not None:
                    setattr(big, name, torch.zeros_like(old))
                    del old
            torch.cuda.synchronize(dev)
            torch.cuda.empty_cache()
            big.reset_state()
        big.run(0, 6)
        # batches of 100 launches, the MEDIAN batch (single passes at this size carry a hiccup of 10-40 % now and then:
        # profiles/r05/ab_variants.txt section 6)
        sm = event_timed(big, lambda t0_, t1_: big.run(t0_, t1_, join=False), 0, n_s, 100, 5 if k == 0 else 3,
                         lanes=big.per_step_stream_list()) / 100
        per_placement.append((float(np.median(sm)), float(sm.min()), float(sm.max())))
    del spacers
    order = sorted(range(len(per_placement)), key=lambda i: per_placement[i][0])
    sm_med, sm_min, sm_max = per_placement[order[len(order) // 2]]         # the median placement (upper median of an even count)
    ach = Ab * n_big / sm_med / 1e9
    pools_c = (ctypes.c_int32 * G)(*big.pools)
    streamed = [bool(big.lib.fiveeq_rows_streamed(G, pools_c, n_, n_big, w)) for _, n_, _ in big.per_step_launches()]
    roofline["hbm_resident"] = {"members": n_big, "state_and_parameter_bytes": resident,
                                "x_infinity_cache": resident / INFINITY_CACHE, "avg_launch_us": sm_med * 1e6,
                                "batch_us_min_median_max": [sm_min * 1e6, sm_med * 1e6, sm_max * 1e6],
                                "achieved": ach, "frac": ach / HBM_PEAK_GBS, "chunk_major": False,
                                "regime": regime(resident),
                                "placements": {"frac_each": [Ab * n_big / v[0] / 1e9 / HBM_PEAK_GBS for v in per_placement],
                                               "is": "the same engine with its stored-trajectory buffers (C, T) re-allocated: [0] = as "
                                                     "constructed; `frac` = the median placement (profiles/r06/hbm_low_mode.txt)"},
                                "rows": "streamed (non-temporal)" if all(streamed) else "cached",
                                "frac_of_best_copy": ach / best_copy_gbs,
                                "concurrent_launches": big.per_step_streams,
                                "algorithmic_bytes_per_step": Ab * n_big}
    roofline["hbm_resident_frac"] = ach / HBM_PEAK_GBS
    big.close()


def ordered(roofline):
    """The scalars a reader needs come FIRST in the object (the driver's parser keeps the head of a nested object)."""
    return {**{k: roofline[k] for k in ROOFLINE_FIRST if k in roofline}, **{k: v for k, v in roofline.items() if k not in ROOFLINE_FIRST}}
