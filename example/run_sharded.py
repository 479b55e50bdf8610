#!/usr/bin/env python3
"""A whole ensemble job over the GPUs of one node: BASELINE configs[3] / [4] in 60 lines.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        example/run_sharded.py --members 10000000 [--dtype f32] [--emissions RCP45_EMISSIONS.csv] [--out summary.csv]

One rank per GPU.  Every rank draws ONLY its contiguous shard of the Latin hypercube on its own device, advances it with no
communication at all (members never interact), and the job ends with ONE exchange over RCCL: per-year moments on every rank,
exact percentiles (selection, not a sort of gathered rows) on rank 0, which writes them as CSV.  `--backend gloo` rehearses the
same job with several ranks sharing one card (the exchange then goes through host memory)."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fiveeqscm_amd import emissions, hostbind, params, scenario  # noqa: E402
from fiveeqscm_amd.distributed import shard_bounds  # noqa: E402
from fiveeqscm_amd.engine import EnsembleEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", type=int, default=10_000_000, help="members of the WHOLE ensemble")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--emissions", default=None, help="CSV in RCP column layout (default: the synthetic RCP-like scenario)")
    ap.add_argument("--years", default="249,499,749", help="step indices whose T is summarised")
    ap.add_argument("--mode", default="auto", help="launch form (engine.run): auto, per_step, fused, ...")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--out", default=None, help="rank 0 writes the summary here (CSV); default: stdout")
    a = ap.parse_args()
    for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
        os.environ.setdefault(key, val)                            # started without a launcher: a one-rank job
    rank, world, local = (int(os.environ[k]) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
    # this rank's host thread next to its GPU (sysfs, applied before anything touches the GPU; no wrapper, no re-exec)
    try:
        hostbind.bind_rank(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    except Exception:  # noqa: BLE001 - an optimisation: never a reason not to run
        pass
    dev = torch.device(f"cuda:{local % torch.cuda.device_count()}")
    torch.cuda.set_device(dev)
    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)          # RCCL on ROCm
    else:
        dist.init_process_group("gloo")
    try:
        if a.emissions:
            years, E = scenario.read_emissions_csv(a.emissions)
        else:
            E = emissions.rcp_like_emissions(750, 3)
            years = np.arange(1765, 1765 + E.shape[0])
        steps = [int(s) for s in a.years.split(",") if int(s) < E.shape[0]]
        dtype = torch.float64 if a.dtype == "f64" else torch.float32
        lo, hi = shard_bounds(a.members, rank, world)             # contiguous member ranges, the same design for any world size
        p = params.sample_ensemble_shard(params.default_params("multigas"), a.members, lo, hi, device=dev, dtype=dtype)
        eng = EnsembleEngine(p, hi - lo, E, dtype=dtype, device=dev, output_steps=steps, store_concentrations=False,
                             collect_stats=True)
        eng.run(mode=a.mode)                                      # no collective while stepping
        summ = eng.gather_summary(steps)                          # THE exchange: moments everywhere, percentiles on rank 0
        if rank == 0:
            scenario.write_summary_csv(a.out or "/dev/stdout", np.asarray(years)[steps], summ)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
