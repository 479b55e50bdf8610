#!/usr/bin/env python3
"""Where the rows of an ensemble LIE decides the beyond-the-cache rate of the per-step kernel (round 6: tools/hbm_low_mode.py
reproduces the `hbm_resident` low mode inside ONE process by re-allocating the engine).  This probe takes the allocator out of
the picture: one arena, the engine's row tensors (r [3G, N], q [2, N], R [SP, N], S [2, N]; optionally C / T) re-seated at
chosen byte offsets inside it, the 8M-member one-launch-per-step rate measured for each placement.

  random   : `--trials` random placements (offsets = multiples of --granule bytes), all offsets logged with the rate;
  realloc  : the engine's OWN tensors, one of them (round-robin over r, q, R, S, C, T) re-allocated through torch per trial (behind
             a random-sized spacer, the old one freed afterwards): whose re-allocation moves the rate?
  shift    : the default back-to-back placement with ONE tensor (--which r|q|R|S) moved by k x --step bytes, k = 0 .. --trials-1:
             the period and the depth of the interference, if it has a structure.

One JSON line per placement."""
import argparse
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

NAMES = ("r", "q", "R", "S")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["random", "shift", "realloc"])
    ap.add_argument("--members", type=int, default=8_000_000)
    ap.add_argument("--trials", type=int, default=32)
    ap.add_argument("--granule", type=int, default=4096)
    ap.add_argument("--which", default="R")
    ap.add_argument("--step", type=int, default=4096)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--batches", type=int, default=3)
    ap.add_argument("--no-trajectory", action="store_true")
    ap.add_argument("--only", default="r,q,R,S,C,T", help="realloc: the tensors re-allocated in turn")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    N, G, n_s = a.members, 3, 112
    p1 = params.sample_ensemble_shard(params.default_params("multigas"), 1_000_000, device=dev)
    reps = -(-N // 1_000_000)
    pb = dict(p1)
    for key in ("r0", "rC", "rT", "q"):
        pb[key] = p1[key].repeat(1, reps)[:, :N].contiguous()
    E = emissions.rcp_like_emissions(750, G)
    eng = EnsembleEngine(pb, N, E[250:250 + n_s], device=dev, chunk_members=0, per_step_streams=a.streams,
                         store_trajectory=not a.no_trajectory)
    warm = EnsembleEngine(p1, 1_000_000, E[:40], device=dev, store_trajectory=False)
    legs.spin_up(warm, dev)
    warm.close()
    orig = {k: getattr(eng, k).clone() for k in NAMES}
    rows = {k: orig[k].shape[0] for k in NAMES}
    nbytes = {k: rows[k] * N * 8 for k in NAMES}
    span = sum(nbytes.values())
    slack = 256 << 20
    arena = torch.empty(span + 4 * slack + (4 << 20), dtype=torch.uint8, device=dev)
    base = (-arena.data_ptr()) % (2 << 20)                               # 2 MiB-aligned start inside the arena
    rng = np.random.default_rng(12345)

    def seat(offsets):
        for k in NAMES:
            view = arena[base + offsets[k]: base + offsets[k] + nbytes[k]].view(torch.float64).view(rows[k], N)
            view.copy_(orig[k])
            setattr(eng, k, view)
        eng.reset_state()

    def packed():
        off, cur = {}, 0
        for k in NAMES:
            off[k] = cur
            cur += -(-nbytes[k] // (2 << 20)) * (2 << 20)                # each tensor on its own 2 MiB boundary
        return off

    def measure():
        eng.run(0, 6)
        torch.cuda.synchronize()
        sm = [float(legs.event_timed(eng, lambda t0_, t1_: eng.run(t0_, t1_, join=False), 0, n_s, 100, 1,
                                     lanes=eng.per_step_stream_list())[0]) / 100 for _ in range(a.batches)]
        return float(np.median(sm))

    Ab = eng.bytes_per_member_step("per_step")
    if a.mode == "realloc":
        keep = []
        orig_src = torch.empty(1 << 26, dtype=torch.float64, device=dev).normal_()
        order = a.only.split(",")
        for trial in range(a.trials):
            which = None if trial == 0 else order[(trial - 1) % len(order)]
            if which is not None:
                old = getattr(eng, which)
                keep.append(torch.empty(int(rng.integers(1, 2048)) << 20, dtype=torch.uint8, device=dev))      # shifts the next allocation
                new = torch.empty_like(old)
                new.copy_(old)
                setattr(eng, which, new)
                del old
                if len(keep) > 6:
                    keep.pop(0)
                torch.cuda.synchronize()
                torch.cuda.empty_cache()                                # the freed block goes back to the driver, not to the next trial
            eng.reset_state()
            dt = measure()
            # does a PLAIN streaming write into the same buffers see the placement too?  (non-temporal copy of 512 MiB from a fixed
            # source into the start, the middle and the end of C and into T; GB/s written)
            wr = {}
            if eng.C is not None:
                import ctypes
                n_el = 1 << 26
                src = orig_src
                for name, buf in (("C", eng.C), ("T", eng.T)):
                    flat = buf.view(-1)
                    rates = []
                    for start in (0, (flat.numel() // 2) // 1024 * 1024, (flat.numel() - n_el) // 1024 * 1024):
                        dst_ptr = flat.data_ptr() + start * 8
                        call = lambda: eng.lib.fiveeq_stream_copy_nt_f64(n_el, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst_ptr), eng._stream())  # noqa: E731
                        call()
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(4):
                            call()
                        e1.record()
                        e1.synchronize()
                        rates.append(round(4 * n_el * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1))
                    wr[name] = rates
            print(json.dumps({"mode": a.mode, "trial": trial, "which": which, "us_per_step": dt * 1e6,
                              "frac_of_8TBs": Ab * N / dt / 8e12, "nt_copy_into_GBs_written": wr,
                              "ptrs": {k: hex(getattr(eng, k).data_ptr()) for k in ("r", "q", "R", "S", "C", "T")}}), flush=True)
        return
    for trial in range(a.trials):
        off = packed()
        if a.mode == "random":
            if trial > 0:                                               # trial 0 = the packed placement itself
                order = list(rng.permutation(len(NAMES)))
                cur = 0
                for i in order:
                    k = NAMES[i]
                    cur += int(rng.integers(0, slack // a.granule)) * a.granule
                    off[k] = cur
                    cur += nbytes[k]
        else:
            off[a.which] += trial * a.step
            for k in NAMES[NAMES.index(a.which) + 1:]:                  # keep the later tensors clear of the moved one
                off[k] += slack
        seat(off)
        dt = measure()
        print(json.dumps({"mode": a.mode, "trial": trial, "which": a.which if a.mode == "shift" else None,
                          "us_per_step": dt * 1e6, "frac_of_8TBs": Ab * N / dt / 8e12,
                          "offsets": {k: int(off[k]) for k in NAMES}, "base_mod_1GiB": int((arena.data_ptr() + base) % (1 << 30)),
                          "ptrs": {k: hex(getattr(eng, k).data_ptr()) for k in NAMES},
                          "C_ptr": None if eng.C is None else hex(eng.C.data_ptr())}), flush=True)


if __name__ == "__main__":
    main()
