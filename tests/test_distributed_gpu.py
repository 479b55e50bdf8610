"""End-to-end multi-process run on real kernels: two ranks (sharing the one GPU of the test box, gloo
for the exchange — the RCCL path differs only in the backend string) each advance their member shard
through the C ABI; the summary exchange must reproduce a single-process run of the whole ensemble."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

N_TOTAL, N_STEPS = 20_000 + 13, 120
YEARS = [40, 119]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from fiveeqscm_amd import emissions, params
    from fiveeqscm_amd.distributed import gather_summary, histogram_percentiles, reduce_stats, shard_bounds
    from fiveeqscm_amd.engine import EnsembleEngine
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = params.sample_ensemble(params.default_params("multigas"), N_TOTAL)
        lo, hi = shard_bounds(N_TOTAL, rank, world)
        p = dict(full)
        for k in ("r0", "rC", "rT", "q"):
            p[k] = np.ascontiguousarray(full[k][:, lo:hi])
        E = emissions.rcp_like_emissions(N_STEPS, 3)
        eng = EnsembleEngine(p, hi - lo, E, device="cuda:0", output_steps=YEARS, collect_stats=True)
        eng.run(mode="fused" if rank else "per_step")           # the two paths are bit-identical
        torch.cuda.synchronize()
        summ = gather_summary(eng.T, percentiles=(5.0, 50.0, 95.0))
        mom = reduce_stats(eng.stats_sums())
        hp, tot = histogram_percentiles(eng.T_histogram(-1.0, 6.0, 4096), -1.0, 6.0, (5.0, 50.0, 95.0))
        if rank == 0:
            ref = EnsembleEngine(full, N_TOTAL, E, device="cuda:0", collect_stats=True)
            ref.run()
            torch.cuda.synchronize()
            T = ref.T.cpu().numpy()
            want = np.percentile(T[YEARS], (5.0, 50.0, 95.0), axis=1).T
            ok = (np.allclose(summ["percentiles"].numpy(), want, rtol=1e-13)
                  and np.allclose(summ["mean"].numpy(), T[YEARS].mean(1), rtol=1e-13)
                  and np.allclose(mom["mean"].numpy(), T.mean(1), rtol=1e-12, atol=1e-15)
                  and np.allclose(mom["var"].numpy(), T.var(1), rtol=1e-8, atol=1e-16)
                  and np.array_equal(mom["min"].numpy(), T.min(1)) and np.array_equal(mom["max"].numpy(), T.max(1))
                  and mom["count"].tolist() == [float(N_TOTAL)] * N_STEPS
                  and tot.tolist() == [float(N_TOTAL)] * 2
                  and np.abs(hp.numpy() - want).max() < 7.0 / 4096)
            q.put(bool(ok))
        else:
            q.put(summ["percentiles"] is None and mom["count"][0].item() == float(N_TOTAL))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_match_single_process():
    assert torch.cuda.is_available()
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(results)
