"""The shard-computable Latin hypercube (params.lhs_rows / sample_ensemble_shard; C ABI fiveeq_lhs_rows_*):
the design of the multi-GPU runs.  CPU tests: the NumPy twin, the library's host
entry point (same C++ functions the kernel runs), stratification, and shard invariance — a rank computes
only its members and the union over any partition is the single-shard design bit for bit.  The kernel
itself is compared with the NumPy twin in tests/test_engine_gpu.py."""
import ctypes

import numpy as np
import pytest

from fiveeqscm_amd import _capi
from fiveeqscm_amd import params as prm
from fiveeqscm_amd.distributed import shard_bounds


@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 17, 256, 257, 1000, 4096, 65537, 100_003])
def test_one_member_per_stratum_in_every_dimension(N):
    u = prm.lhs_rows(N, [0, 1, 7, 10])
    assert u.shape == (4, N) and u.min() > 0.0 and u.max() < 1.0
    for row in u:
        assert np.array_equal(np.sort(np.floor(row * N).astype(np.int64)), np.arange(N))
    if N > 1000:            # dimensions are different permutations, not copies of one
        assert abs(np.corrcoef(u)[0, 1]) < 0.05


def test_permutation_is_a_bijection_and_keyed():
    N = 12345
    m = np.arange(N, dtype=np.uint64)
    a = prm.lhs_permute(m, N, prm._lhs_dim_key(prm.LHS_SEED, 0))
    b = prm.lhs_permute(m, N, prm._lhs_dim_key(prm.LHS_SEED, 1))
    c = prm.lhs_permute(m, N, prm._lhs_dim_key(prm.LHS_SEED + 1, 0))
    for x in (a, b, c):
        assert np.array_equal(np.sort(x), m)
    assert (a != b).mean() > 0.99 and (a != c).mean() > 0.99 and (a != m).mean() > 0.99


@pytest.mark.parametrize("world", [2, 3, 8])
def test_union_of_shards_is_the_single_shard_design(world):
    N = 10_007                                   # ragged split
    base = prm.default_params("multigas")
    full = prm.sample_ensemble_shard(base, N)
    parts = [prm.sample_ensemble_shard(base, N, *shard_bounds(N, r, world)) for r in range(world)]
    for name in ("r0", "rC", "rT", "q"):
        assert np.array_equal(np.concatenate([p[name] for p in parts], axis=1), full[name]), name
    assert np.array_equal(np.concatenate([p["TCR"] for p in parts]), full["TCR"])


def test_rank_3_of_8_slice_of_the_10M_design_needs_only_its_shard():
    """BASELINE configs[3]: rank 3 of 8 computes members [3.75M, 5M) of the 10M-member design without
    ever touching the other 8.75M; spot-check a window against the same window computed alone."""
    N = 10_000_000
    lo, hi = shard_bounds(N, 3, 8)
    assert (lo, hi) == (3_750_000, 5_000_000)
    u = prm.lhs_rows(N, range(11), lo, lo + 4096)
    v = prm.lhs_rows(N, range(11), lo + 1000, lo + 1100)
    assert np.array_equal(u[:, 1000:1100], v)
    assert np.all((u > 0) & (u < 1))


def test_library_host_twin_equals_numpy_twin():
    lib = _capi.load()
    for N, lo, hi in ((1, 0, 1), (1000, 0, 1000), (12345, 100, 9000), (10_000_000, 3_750_000, 3_760_000),
                      (100_000_000, 99_990_000, 100_000_000)):
        out = np.full((11, hi - lo + 3), -1.0)
        rc = lib.fiveeq_lhs_rows_host_f64(prm.LHS_SEED, N, lo, hi - lo, 0, 11, hi - lo + 3,
                                          out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        assert np.array_equal(out[:, :hi - lo], prm.lhs_rows(N, range(11), lo, hi)), N
        assert np.all(out[:, hi - lo:] == -1.0)                       # ld padding untouched
    out = np.empty((1, 4))
    ptr = out.ctypes.data_as(ctypes.c_void_p)
    assert lib.fiveeq_lhs_rows_host_f64(1, 10, 8, 4, 0, 1, 4, ptr) == _capi.E_INVALID      # [8,12) outside [0,10)
    assert lib.fiveeq_lhs_rows_host_f64(1, 0, 0, 0, 0, 1, 4, ptr) == _capi.E_INVALID
    assert lib.fiveeq_lhs_rows_host_f64(1, 10, 0, 4, 0, 1, 3, ptr) == _capi.E_INVALID       # ld < n
    assert lib.fiveeq_lhs_rows_f64(1, 10, 0, 4, 0, 1, 3, ptr, None) == _capi.E_INVALID      # validated before any launch


def test_sampling_rule_ranges_and_k_q():
    N = 4000
    base = prm.default_params("multigas")
    p = prm.sample_ensemble_shard(base, N)
    u = (p["r0"][0] / base["r0"][0] - 0.8) / 0.4
    assert np.array_equal(np.sort(np.floor(u * N + 1e-9).astype(int).clip(0, N - 1)), np.arange(N))
    tcr, ecs = p["TCR"], p["ECS"]
    assert np.all(ecs >= 1.1 * tcr - 1e-12) and tcr.min() >= 1.0 and ecs.max() <= 4.5 and np.all(p["q"] > 0)
    np.testing.assert_allclose(p["q"], prm.k_q(tcr, ecs, base["d"], prm.forcing_2x(base)), rtol=1e-13)
    assert np.all(p["rC"][1] == 0.0)                                   # CH4 has no rC term: stays exactly zero
    assert np.array_equal(p["r0"], prm.sample_ensemble_shard(base, N)["r0"])                   # deterministic
    assert not np.array_equal(p["r0"], prm.sample_ensemble_shard(base, N, seed=7)["r0"])       # seeded
