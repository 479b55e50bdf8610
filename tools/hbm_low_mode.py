#!/usr/bin/env python3
"""The `hbm_resident` low mode: the per-step kernel on an ensemble several times the
Infinity Cache runs at 0.79-0.80 of 8 TB/s in most processes and at 0.70-0.72 in some (4 of ~30 in round 5), for the whole life
of the process.  This tool measures, in ONE process and as often as asked:

  * the 8M-member beyond-the-cache rate exactly as bench.py's hbm_resident leg does (100-launch HIP-event batches, chunk-major
    off, trajectories stored), once per CYCLE — a cycle re-creates the engine (fresh allocations behind a random-sized spacer),
    so a mode tied to where the allocator put the rows shows up INSIDE one process;
  * optionally (--config3) the cache-resident 1M-member rate of the default bench line, to see whether `value` shares the mode;
  * the clocks and the power the driver reports (sysfs pp_dpm_*, hwmon) right after each batch — no child process, so the tool
    can run directly under `rocprofv3 --pmc ... -- python3 tools/hbm_low_mode.py ...` (FIVEEQ_SIDE_STREAM_PROBE=0 there).

One JSON line per cycle on stdout.  --tag goes into every line (the shell loop's run index)."""
import argparse
import glob
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import legs  # noqa: E402
from fiveeqscm_amd import emissions, params  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402


def _read(path):
    try:
        with open(path) as fh:
            return fh.read()
    except OSError:
        return None


def driver_state():
    """{sclk, mclk, fclk: MHz of the level the driver marks current ('*'); power_w} from sysfs; None where unreadable."""
    out = {}
    for name in ("sclk", "mclk", "fclk", "socclk"):
        val = None
        for path in glob.glob(f"/sys/class/drm/card*/device/pp_dpm_{name}"):
            text = _read(path)
            if text:
                cur = [ln for ln in text.splitlines() if ln.rstrip().endswith("*")]
                if cur:
                    try:
                        val = int("".join(c for c in cur[0].split(":")[1] if c.isdigit()))
                    except (IndexError, ValueError):
                        val = cur[0].strip()
        out[name] = val
    power = None
    for path in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"):
        text = _read(path)
        if text:
            try:
                power = int(text) / 1e6
            except ValueError:
                pass
    out["power_w"] = power
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=8_000_000)
    ap.add_argument("--cycles", type=int, default=1)
    ap.add_argument("--batches", type=int, default=9)
    ap.add_argument("--launches", type=int, default=100)
    ap.add_argument("--config3", action="store_true")
    ap.add_argument("--streams", default="auto")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    G, n_s = 3, 112
    rng = np.random.default_rng(os.getpid())
    p1 = params.sample_ensemble_shard(params.default_params("multigas"), 1_000_000, device=dev)
    reps = -(-a.members // 1_000_000)
    pb = dict(p1)
    for key in ("r0", "rC", "rT", "q"):
        pb[key] = p1[key].repeat(1, reps)[:, :a.members].contiguous()
    E = emissions.rcp_like_emissions(750, G)
    small = None
    if a.config3:
        small = EnsembleEngine(p1, 1_000_000, E, device=dev)
    warm = EnsembleEngine(p1, 1_000_000, E[:40], device=dev, store_trajectory=False)
    legs.spin_up(warm, dev)
    warm.close()
    spacer = None
    for cyc in range(a.cycles):
        big = EnsembleEngine(pb, a.members, E[250:250 + n_s], device=dev, chunk_members=0,
                             per_step_streams=a.streams if a.streams == "auto" else int(a.streams))
        big.run(0, 6)
        torch.cuda.synchronize()
        states = []
        sm = []
        for _ in range(a.batches):
            sm.append(float(legs.event_timed(big, lambda t0_, t1_: big.run(t0_, t1_, join=False), 0, n_s, a.launches, 1,
                                             lanes=big.per_step_stream_list())[0]) / a.launches)
            states.append(driver_state())
        sm = np.array(sm)
        Ab = big.bytes_per_member_step("per_step")
        line = {"tag": a.tag, "pid": os.getpid(), "cycle": cyc, "members": a.members, "streams": big.per_step_streams,
                "us_per_launch_median": float(np.median(sm)) * 1e6, "us_min": float(sm.min()) * 1e6, "us_max": float(sm.max()) * 1e6,
                "frac_of_8TBs": Ab * a.members / float(np.median(sm)) / 8e12,
                "frac_batches": [round(Ab * a.members / float(v) / 8e12, 4) for v in sm],
                "R_ptr": hex(big.R.data_ptr()), "C_ptr": hex(big.C.data_ptr()),
                "sclk": [s["sclk"] for s in states][-1], "mclk": [s["mclk"] for s in states][-1],
                "fclk": [s["fclk"] for s in states][-1], "power_w": [s["power_w"] for s in states][-1],
                "side_streams": big.side_stream_report()}
        if small is not None:
            blocks = legs.event_timed(small, lambda t0_, t1_: small.run(t0_, t1_, join=False), 0, 750, 100, 15,
                                      lanes=small.per_step_stream_list()) / 100
            line["config3_us_per_step_median"] = float(np.median(blocks)) * 1e6
            line["config3_frac_of_8TBs"] = 248.0 * 1_000_000 / float(np.median(blocks)) / 8e12
        print(json.dumps(line), flush=True)
        big.close()
        del big
        if a.cycles > 1:
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            spacer = torch.empty(int(rng.integers(1, 4096)) << 20, dtype=torch.uint8, device=dev)     # shifts the next allocations
    del spacer


if __name__ == "__main__":
    main()
