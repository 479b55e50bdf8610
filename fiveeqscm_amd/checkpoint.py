"""Checkpoint / resume of an EnsembleEngine: the state a resumed run needs, and what the run has
reduced or stored so far, as host NumPy arrays."""
import numpy as np
import torch


class CheckpointMixin:
    """state_dict() / load_state_dict() of fiveeqscm_amd.engine.EnsembleEngine (which supplies every attribute used here)."""

    def state_dict(self, include_outputs="summaries"):
        """Checkpoint: everything a resumed run needs besides the (immutable) parameters and drive
        table — pools, thermal boxes, in inverse mode the per-member cumulative emissions, the index
        `t_next` of the first step not yet run — as host NumPy arrays (state in fp64), plus, by `include_outputs`:
          "summaries" (default)  what the run has REDUCED so far: `T_hist` ([n_steps, n_bins] int64, 24 MB at 750 x 4096)
                                 and the per-step moment sums of steps [0, t_next) folded to [n_steps, 5] (30 KB) — small
                                 whatever the ensemble size;
          True                   also the raw buffers: per-wave records `T_stats` (0.5 B per member-step: 4.7 GB at
                                 12.5M members x 750 steps) and the stored C/T rows ((G+1) w bytes per member and stored
                                 step: 24 GB for 1M fp64 members x 750 steps) — sized like the run, so opt-in;
          False                  the state only.
        Resume with `load_state_dict` and `run(state["t_next"], ...)`: bit-identical to an uninterrupted run
        (checkpoint/resume)."""
        if include_outputs not in (True, False, "summaries"):
            raise ValueError("include_outputs must be True, False or 'summaries'")
        torch.cuda.synchronize(self.device)
        out = {"R": self.R.double().cpu().numpy(), "S": self.S.double().cpu().numpy(), "t_next": int(self.t_next)}
        if self.cumE is not None:
            out["cumE"] = self.cumE.double().cpu().numpy()
        if include_outputs:
            if self.T_hist is not None:
                out["T_hist"] = self.T_hist.cpu().numpy()
            if self.collect_stats:
                # only the steps this engine HAS moments for (it ran them, or a checkpoint brought them): a run that began
                # at t_begin > 0, or a state-only checkpoint loaded before it, leaves the earlier steps out — their
                # zero-filled records are not moments
                sums = np.zeros((self.n_steps, 5), dtype=np.float64)
                valid = self._stats_have.copy()
                if valid.any():
                    lo_t, hi_t = int(np.nonzero(valid)[0][0]), int(np.nonzero(valid)[0][-1]) + 1
                    sums[lo_t:hi_t] = self.stats_sums(lo_t, hi_t).cpu().numpy()
                    sums[~valid] = 0.0
                out["_step_sums"], out["_step_sums_valid"] = sums, valid
        if include_outputs is True:
            for name in ("T_stats", "C", "T"):
                buf = getattr(self, name)
                if buf is not None:
                    out[name] = buf.cpu().numpy()
        return out

    def load_state_dict(self, state):
        """Restore a checkpoint (after `join()` if a run(..., join=False) is still outstanding).  One WITHOUT summaries
        (include_outputs=False) restores the state only: the accumulators of this engine (T_hist, per-step moments) are then
        cleared, because they describe a run this state is not from."""
        # VALIDATE everything first, then mutate: a checkpoint whose n_steps / n_bins / n_rows do not match raises with the
        # engine exactly as it was (round 5 copied R, S and reset the masks before it looked at the accumulators' shapes)
        state_names = ("R", "S") + (("cumE",) if self.cumE is not None else ())
        if "T_stats" in state and self.collect_stats:
            self._wave_stats()                                   # the checkpoint carries wave records: make room for them (no state touched)
        staged = {}
        for name in state_names + ("T_stats", "T_hist", "C", "T", "_step_sums"):
            dst = getattr(self, name)
            if name in state_names and name not in state:
                raise KeyError(f"checkpoint has no {name!r}")
            if dst is None or name not in state:
                continue
            src = np.asarray(state[name], dtype=np.float64) if name in state_names else np.asarray(state[name])
            if src.shape != tuple(dst.shape):
                raise ValueError(f"{name}: checkpoint shape {src.shape}, engine {tuple(dst.shape)}")
            if name == "T_hist" and src.dtype.kind not in "iu":
                raise ValueError(f"T_hist: checkpoint dtype {src.dtype}, want integer counts")
            if name != "T_hist" and src.dtype.kind not in "fiu":
                raise ValueError(f"{name}: checkpoint dtype {src.dtype}, want numbers")
            staged[name] = src
        valid = None
        if "_step_sums_valid" in state and self._step_sums is not None:
            valid = np.asarray(state["_step_sums_valid"], dtype=bool)
            if valid.shape != self._step_sums_valid.shape:
                raise ValueError(f"_step_sums_valid: checkpoint shape {valid.shape}, engine {self._step_sums_valid.shape}")
            if valid.any() and "_step_sums" not in staged and "T_stats" not in staged:
                raise ValueError("checkpoint marks step moments valid but carries neither _step_sums nor T_stats")
        t_next = int(state.get("t_next", 0))
        if not 0 <= t_next <= self.n_steps:
            raise ValueError(f"t_next={t_next} outside 0..{self.n_steps}")

        if self._ps_unjoined:
            self.join()
        for name, src in staged.items():                         # the accumulators' contents BEFORE the masks that describe them
            dst = getattr(self, name)
            dst.copy_(torch.from_numpy(src).to(dst.dtype))
        if self.T_hist is not None and "T_hist" not in staged:
            self.T_hist.zero_()
        self.t_next = t_next
        self._step_sums_valid[:] = False
        self._stats_have[:] = False
        if valid is not None:
            # the saver's own mask of the steps it had moments for: the only word on which records are moments (a saver that
            # began at t_begin > 0 holds zero-filled records before it)
            self._step_sums_valid[:] = valid & ("_step_sums" in staged)
            self._stats_have[:] = valid
        elif "T_stats" in staged:                              # a checkpoint from before the mask existed: raw wave records of
            self._stats_have[:self.t_next] = True              # a run from step 0
