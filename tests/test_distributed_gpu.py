"""End-to-end multi-process runs on real kernels.

(1) Four ranks sharing the one GPU of the test box (gloo for the exchange; the process guard of the GPU pool allows
    at most 6 processes on a card, so the 8-rank rehearsal of BASELINE configs[3] is the CPU test in
    tests/test_distributed.py): each rank draws ONLY its shard of the shard-computable Latin hypercube on the
    device, advances it through the C ABI, and the summary exchange must reproduce np.percentile / mean of the
    ORACLE's T over the whole ensemble, plus the engine's own single-process results bit for bit.
(2) The same exchange over RCCL ("nccl"), one rank per GPU: skips itself on a box with fewer than two GPUs, so the
    first multi-GPU lease exercises dist.gather / all_reduce over xGMI in a test rather than in bench.py.
(3) RCCL first contact on ONE GPU: a "nccl" process group of one rank with distributed.force_collectives, so every
    collective the summary makes (all_gather fp64, all_reduce SUM int64 [750, 4096], MIN / MAX fp64, gather of a device
    tensor with a receive list, barrier) executes on the device backend instead of being skipped by "one rank".
"""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

N_TOTAL, N_STEPS = 60_000 + 13, 120
YEARS = [40, 119]
PCT = (5.0, 50.0, 95.0)


def _worker(rank, world, port, backend, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from fiveeqscm_amd import emissions, params
    from fiveeqscm_amd.distributed import gather_summary, histogram_percentiles, reduce_stats, shard_bounds
    from fiveeqscm_amd.engine import EnsembleEngine
    dev_index = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        base = params.default_params("multigas")
        lo, hi = shard_bounds(N_TOTAL, rank, world)
        p = params.sample_ensemble_shard(base, N_TOTAL, lo, hi, device=dev)        # O(shard), drawn on the device
        E = emissions.rcp_like_emissions(750, 3)[180:180 + N_STEPS]
        eng = EnsembleEngine(p, hi - lo, E, device=dev, output_steps=YEARS, store_concentrations=False,
                             collect_stats=True, hist=(-1.0, 6.0, 4096))
        eng.run(mode=("per_step", "fused")[rank % 2])                              # the paths are bit-identical
        torch.cuda.synchronize()
        st = {}
        summ = gather_summary(eng.T, percentiles=PCT, stats=st)
        mom = reduce_stats(eng.stats_sums())
        hp, tot = histogram_percentiles(eng.T_hist, -1.0, 6.0, PCT)                # in-loop histograms, all-reduced
        if rank == 0:
            from oracle import c_oracle
            whole = params.sample_ensemble_shard(base, N_TOTAL)                    # host twin of the same design
            T = c_oracle.run(E, whole, N_TOTAL, keep=("T",), n_threads=8)["T"]     # the ORACLE, in one piece
            want = np.percentile(T, PCT, axis=1).T
            got = summ["percentiles"].cpu().numpy()
            ok = {
                "percentiles": np.allclose(got, want[YEARS], rtol=1e-10, atol=1e-13),
                "mean": np.allclose(summ["mean"].cpu().numpy(), T[YEARS].mean(1), rtol=1e-10),
                "moments": (np.allclose(mom["mean"].cpu().numpy(), T.mean(1), rtol=1e-10, atol=1e-13)
                            and np.allclose(mom["var"].cpu().numpy(), T.var(1), rtol=1e-8, atol=1e-16)
                            and np.allclose(mom["min"].cpu().numpy(), T.min(1), rtol=1e-10, atol=1e-13)
                            and np.allclose(mom["max"].cpu().numpy(), T.max(1), rtol=1e-10, atol=1e-13)),
                "count": mom["count"].tolist() == [float(N_TOTAL)] * N_STEPS and tot.tolist() == [float(N_TOTAL)] * N_STEPS,
                "hist": np.abs(hp.cpu().numpy() - want).max() < 2 * 7.0 / 4096,
                "payload": st["bytes_to_root"] < 0.05 * len(YEARS) * N_TOTAL * 8,
            }
            # and the engine's own single-process run of the whole ensemble: bit for bit
            ref = EnsembleEngine(whole, N_TOTAL, E, device=dev, output_steps=YEARS, store_concentrations=False)
            ref.run()
            torch.cuda.synchronize()
            want_e = np.percentile(ref.T.cpu().numpy(), PCT, axis=1).T
            ok["engine"] = np.allclose(got, want_e, rtol=1e-13)
            q.put({k: bool(v) for k, v in ok.items()})
        else:
            q.put({"nonroot": summ["percentiles"] is None and mom["count"][0].item() == float(N_TOTAL)})
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _launch(world, backend):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, backend, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for res in results:
        assert all(res.values()), res


def test_four_ranks_one_gpu_match_the_oracle():
    assert torch.cuda.is_available()
    _launch(4, "gloo")


def test_summary_exchange_over_rccl():
    """One rank per GPU over RCCL (backend "nccl"): all_gather of moments, all_reduce of histograms, gather of the
    percentile candidates to the root.  Needs >= 2 GPUs; the single-GPU test box skips it."""
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"{n} GPU(s) visible: the RCCL leg needs at least 2")
    _launch(min(n, 4), "nccl")


def _first_contact_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from fiveeqscm_amd import distributed as D
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)       # "nccl" is RCCL on ROCm
    calls = []
    real = {name: getattr(dist, name) for name in ("all_reduce", "all_gather", "gather", "barrier")}

    def spy(name):
        def f(*a, **k):
            t = a[0] if a and isinstance(a[0], torch.Tensor) else None
            calls.append((name, None if t is None else (str(t.dtype), t.is_cuda)))
            return real[name](*a, **k)
        return f

    try:
        assert dist.get_backend() == "nccl"
        D.force_collectives(True)
        for name in real:
            setattr(dist, name, spy(name))
        g = torch.Generator(device=dev).manual_seed(7)
        rows = torch.randn((3, 300_001), generator=g, device=dev, dtype=torch.float64) * 0.6 + 1.8
        rows32 = rows.to(torch.float32)
        st = {}
        summ = D.gather_summary(rows, percentiles=PCT, stats=st)
        summ32 = D.gather_summary(rows32, percentiles=PCT)
        hist = torch.randint(0, 1000, (750, 4096), generator=g, device=dev, dtype=torch.int64)
        hp, tot = D.histogram_percentiles(hist, -1.0, 6.0, PCT)
        sums = torch.stack([torch.full((3,), float(rows.shape[1]), dtype=torch.float64, device=dev), rows.sum(1),
                            (rows * rows).sum(1), rows.min(1).values, rows.max(1).values], dim=1)
        mom = D.reduce_stats(sums)
        el = torch.tensor([1.25, 0.5], dtype=torch.float64, device=dev)       # bench.py's max-over-ranks of the block times
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()
        x = rows.cpu().numpy()
        ok = {
            "percentiles": np.allclose(summ["percentiles"].cpu().numpy(), np.percentile(x, PCT, axis=1).T, rtol=1e-13),
            "percentiles_f32": np.allclose(summ32["percentiles"].cpu().numpy(),
                                           np.percentile(rows32.cpu().numpy().astype(np.float64), PCT, axis=1).T, rtol=1e-13),
            "mean": np.allclose(summ["mean"].cpu().numpy(), x.mean(1), rtol=1e-12),
            "var": np.allclose(summ["var"].cpu().numpy(), x.var(1), rtol=1e-9),
            "minmax": np.array_equal(mom["min"].cpu().numpy(), x.min(1)) and np.array_equal(mom["max"].cpu().numpy(), x.max(1)),
            "hist_total": torch.equal(tot.cpu(), hist.sum(1).to(torch.float64).cpu()) and bool(torch.isfinite(hp).all()),
            "elapsed_max": el.tolist() == [1.25, 0.5],
            "payload": st["bytes_to_root"] == 0 and st["allreduce_bytes"] == 3 * 4096 * 8,
            # every collective ran on DEVICE tensors: nothing was staged through the host
            "on_device": all(c[1] is None or c[1][1] for c in calls),
            "calls": ({c[0] for c in calls} == {"all_reduce", "all_gather", "gather", "barrier"}
                      and ("all_reduce", ("torch.int64", True)) in calls and ("gather", ("torch.float64", True)) in calls
                      and ("gather", ("torch.float32", True)) in calls),
        }
        q.put({k: bool(v) for k, v in ok.items()})
    except Exception as exc:  # noqa: BLE001 - report instead of leaving the parent waiting on the queue
        q.put({f"{type(exc).__name__}: {exc}": False})
        raise
    finally:
        for name, fn in real.items():
            setattr(dist, name, fn)
        dist.destroy_process_group()


def test_rccl_first_contact_on_one_gpu():
    """RCCL executes every call of the summary exchange on the one GPU of the test box (a one-rank "nccl" group with
    force_collectives), in a child process so that the communicator's lifetime is that process's."""
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_first_contact_worker, args=(port, q))
    p.start()
    try:
        res = q.get(timeout=600)
    finally:
        p.join(timeout=120)
    assert p.exitcode == 0
    assert all(res.values()), res


def test_the_sharded_example_job_on_three_ranks(tmp_path):
    """example/run_sharded.py — a whole job as a user would launch it (torch.distributed.run, one rank per GPU; here three
    ranks share the card and exchange over gloo): its CSV must hold the percentiles of ONE engine run over the whole ensemble,
    bit for bit (the design and the summary do not depend on the world size)."""
    import subprocess
    import sys
    from fiveeqscm_amd import emissions, params, scenario
    from fiveeqscm_amd.engine import EnsembleEngine
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "summary.csv"
    N = 90_001
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "example", "run_sharded.py"), "--members", str(N), "--backend", "gloo",
           "--years", "100,749", "--out", str(out)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0",
                                                                                    OMP_NUM_THREADS="2"))
    assert res.returncode == 0, res.stderr[-2000:]
    years, cols = scenario.read_summary_csv(str(out))
    assert years.tolist() == [1865.0, 2514.0] and cols["count"].tolist() == [float(N)] * 2
    whole = params.sample_ensemble_shard(params.default_params("multigas"), N, device="cuda:0")
    eng = EnsembleEngine(whole, N, emissions.rcp_like_emissions(750, 3), device="cuda:0", output_steps=[100, 749],
                         store_concentrations=False)
    eng.run(mode="fused")
    want = eng.gather_summary([100, 749])
    assert np.array_equal(cols["percentiles"], want["percentiles"].numpy())
    np.testing.assert_allclose(cols["mean"], want["mean"].numpy(), rtol=1e-13)
