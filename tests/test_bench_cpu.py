"""bench.py pieces that need no GPU: the plain `--gpus N` form starts its own ranks (as a child launcher job, before anything
could touch a GPU) and reports their failure with its exit code."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_gpus_2_starts_a_two_rank_child_job():
    """No GPU here: both ranks must die on bench.py's own `needs an MI355X` assertion — which proves that the plain call
    started the launcher form with two ranks — and the parent must exit non-zero without printing a line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""                       # also on a GPU box: this test is about the launch, not the run
    env["CUDA_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.stderr.count("bench.py needs an MI355X") >= 2, out.stderr[-3000:]
    assert "torch.distributed" in out.stderr or "ChildFailedError" in out.stderr


def test_launcher_environment_is_not_relaunched():
    """With WORLD_SIZE / RANK in the environment (the driver's N > 1 form) the script must NOT start another launcher."""
    sys.path.insert(0, ROOT)
    import importlib
    env_backup = dict(os.environ)
    try:
        os.environ.update(WORLD_SIZE="2", RANK="0")
        bench = importlib.import_module("bench")
        argv = sys.argv
        sys.argv = ["bench.py", "--gpus", "2"]
        try:
            assert bench._self_launch() is None           # returns instead of spawning + sys.exit
        finally:
            sys.argv = argv
    finally:
        os.environ.clear()
        os.environ.update(env_backup)


def test_roofline_regime_labels_and_key_order():
    """benchlib/legs.py pieces that need no GPU: the regime label that goes beside `roofline.frac` (a fraction of the HBM PEAK is
    an HBM fraction only when the bytes cross HBM) and the order of the roofline object (the driver's parser keeps the head)."""
    sys.path.insert(0, ROOT)
    from benchlib import legs
    assert legs.regime(152e6).startswith("infinity-cache-resident (152 MB")
    assert legs.regime(8 * 19 * 8_000_000).startswith("hbm-streamed (1216 MB") and "4.5x" in legs.regime(8 * 19 * 8_000_000)
    assert legs.regime(180e6, n_seq=14).startswith("chunk-major (180 MB per chunk, 0.67 of the Infinity Cache")
    assert legs.regime(230e6).startswith("hbm-streamed")                     # past 0.8 of the cache without chunking
    r = legs.ordered({"note": 1, "kernel": "k", "hbm_resident_frac": 0.8, "regime": "x", "frac": 0.9, "traffic": None, "unit": "GB/s",
                      "peak": 8000.0, "achieved": 7200.0, "bound": "hbm", "single_launch_frac": 0.85, "zzz": 0})
    assert list(r)[:9] == ["bound", "achieved", "peak", "unit", "frac", "regime", "traffic", "hbm_resident_frac", "single_launch_frac"]
    assert list(r)[-2:] == ["note", "zzz"] and len(r) == 12
