"""Control plane of a multi-rank bench run: gloo over 127.0.0.1 — barriers around the clocked region, MAX of the clocked times,
small per-rank records.  Host scalars only; never inside a clocked region.  (The DATA plane — RCCL — is created by bench.py where
it is first used: inside the watchdog-protected summary section.)"""
import os
import sys

import torch


class stdout_to_stderr:
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


class Control:
    """world == 1 without `force`: every method is the identity (no process group exists)."""

    def __init__(self, world, rank, dev, timeout_s, force=False):
        self.world, self.rank, self.dev, self.dist, self.timeout = world, rank, dev, None, None
        if world > 1 or force:
            # TWO process groups.  CONTROL plane (default group): gloo over 127.0.0.1.  DATA plane: RCCL ("nccl" on ROCm) — the
            # end-of-run summary exchange, the only collective that moves ensemble data; its communicator is created by its first
            # collective, which happens AFTER the measurement and under a watchdog.  Whatever RCCL does on first contact across
            # xGMI (an exception, a hang) can therefore cost the line its `summary`, never its measurement.
            from datetime import timedelta

            import torch.distributed as dist
            if force:
                for key, val in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29513"), ("RANK", "0"), ("WORLD_SIZE", "1")):
                    os.environ.setdefault(key, val)
                from fiveeqscm_amd.distributed import force_collectives
                force_collectives(True)
            self.timeout = timedelta(seconds=timeout_s)
            with stdout_to_stderr():          # gloo announces its connections on STDOUT: the one line must stay the only one
                dist.init_process_group("gloo", timeout=self.timeout)
                dist.barrier()
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def sync_all(self):
        torch.cuda.synchronize(self.dev)
        self.barrier()
        torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, values):
        """Element-wise MAX over the ranks of a list of floats."""
        if self.dist is None:
            return [float(v) for v in values]
        tt = torch.tensor(values, dtype=torch.float64)
        self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in tt.tolist()]

    def gather_over_ranks(self, obj):
        """Every rank's (small, picklable) `obj` as a list indexed by rank, on every rank."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out
