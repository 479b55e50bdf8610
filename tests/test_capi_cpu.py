"""C-ABI checks that need no GPU: the library loads, exports every symbol include/fiveeq.h
declares, and rejects bad arguments on the host before anything could reach a device."""
import ctypes
import os
import re

import pytest

from fiveeqscm_amd import _capi
from fiveeqscm_amd import params as prm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    with open(os.path.join(ROOT, "include", "fiveeq.h")) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(fiveeq_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _capi.load()
    names = _header_symbols()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/fiveeq.h but not exported"
    assert sorted(_capi.SIGNATURES) == names, "ctypes binding and header disagree"
    assert lib.fiveeq_abi_version() == _capi.ABI_VERSION
    assert lib.fiveeq_sizeof_model() == ctypes.sizeof(_capi.Model) == 448
    assert [lib.fiveeq_stats_waves(n) for n in (0, 1, 64, 65, 1_000_000)] == [0, 1, 1, 2, 15625]


def test_shipped_library_is_stamped_with_its_sources_and_has_no_experiment_knobs():
    """csrc/Makefile stamps the sha256 of (fiveeq_capi.hip, fiveeq_device.hpp, fiveeq.h) into the library; the binding
    recomputes it from the tree.  The product build carries no experiment flag (timing hooks, occupancy caps, block shapes)."""
    lib = _capi.load()
    assert lib.fiveeq_source_hash().decode() == _capi.source_hash() and len(_capi.source_hash()) == 64
    assert _capi.build_flags(lib) == ""
    with open(os.path.join(ROOT, "fiveeqscm_amd", "csrc", "fiveeq_device.hpp")) as fh:
        text = fh.read()
    assert "wall_clock64" not in text and "s_getreg" not in text, "experiment code belongs in tools/variants/"


def test_a_library_built_from_other_sources_is_refused(tmp_path, monkeypatch):
    """Same binary, but the tree's sources differ from what it was stamped with: load() must raise, not test a stale .so."""
    import shutil
    copy = tmp_path / "libfiveeq_hip.so"
    shutil.copy(_capi.LIB_PATH, copy)
    monkeypatch.setattr(_capi, "source_hash", lambda: "0" * 64)
    with pytest.raises(ImportError, match="built from other sources"):
        _capi.load(str(copy))
    monkeypatch.setenv("FIVEEQ_ALLOW_STALE_LIB", "1")                     # tools/ only: variants of patched sources
    assert _capi.load(str(copy)).fiveeq_abi_version() == _capi.ABI_VERSION


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(ImportError, match="no CPU fallback"):
        _capi.load(str(tmp_path / "libfiveeq_hip.so"))


def test_layout_supported():
    lib = _capi.load()
    ok = lambda *p: lib.fiveeq_layout_supported(len(p), (ctypes.c_int32 * len(p))(*p))  # noqa: E731
    assert ok(4) and ok(1) and ok(4, 1, 1) and ok(4, 4, 4) and ok(1, 1, 1)
    assert not ok(2, 3) and not ok(5) and not ok(0)
    assert lib.fiveeq_layout_supported(4, (ctypes.c_int32 * 4)(1, 1, 1, 1)) == 0
    assert lib.fiveeq_layout_supported(1, None) == 0


def _call_step(lib, model, n=8, ld=8, n_steps=4, t=0, ptr=0x1000):
    p = ctypes.c_void_p(ptr)
    return lib.fiveeq_step_f64(ctypes.byref(model), n, ld, p, n_steps, t, p, p, p, p, None, None, 0, None, None)


def test_validation_rejects_bad_arguments_before_any_launch():
    """Every call below must return on the host with an error code: the (fake) pointers are never
    dereferenced and no kernel is launched, so this runs without a GPU."""
    lib = _capi.load()
    good = prm.make_model(prm.default_params("co2"))
    cases = [
        (dict(n=0), "n_members"),
        (dict(n=8, ld=4), "ld="),
        (dict(t=4), "t=4"),
        (dict(t=-1), "t=-1"),
        (dict(n_steps=0), "t=0"),
        (dict(ptr=0), "NULL device pointer"),
    ]
    for kw, needle in cases:
        rc = _call_step(lib, good, **kw)
        assert rc == _capi.E_INVALID, kw
        assert needle in lib.fiveeq_last_error().decode(), (kw, lib.fiveeq_last_error())

    def broken(edit):
        m = prm.make_model(prm.default_params("co2"))
        edit(m)
        return m

    bad_models = [
        (lambda m: setattr(m, "n_gas", 0), _capi.E_INVALID),
        (lambda m: setattr(m, "n_gas", 4), _capi.E_INVALID),
        (lambda m: setattr(m, "dt", 0.0), _capi.E_INVALID),
        (lambda m: setattr(m, "dt", float("nan")), _capi.E_INVALID),
        (lambda m: m.d.__setitem__(1, -1.0), _capi.E_INVALID),
        (lambda m: setattr(m.gas[0], "n_pools", 5), _capi.E_INVALID),
        (lambda m: m.gas[0].tau.__setitem__(2, 0.0), _capi.E_INVALID),
        (lambda m: setattr(m.gas[0], "g1", 0.0), _capi.E_INVALID),
        (lambda m: setattr(m.gas[0], "C0", 0.0), _capi.E_INVALID),
        (lambda m: setattr(m.gas[0], "emis2conc", float("inf")), _capi.E_INVALID),
        (lambda m: (setattr(m, "n_gas", 2), setattr(m.gas[0], "n_pools", 2), setattr(m.gas[1], "n_pools", 3),
                    setattr(m.gas[1], "g1", 1.0), setattr(m.gas[1], "C0", 1.0), setattr(m.gas[1], "emis2conc", 1.0),
                    [m.gas[1].tau.__setitem__(i, 1.0) for i in range(3)]), _capi.E_UNSUPPORTED),
    ]
    for edit, want in bad_models:
        assert _call_step(lib, broken(edit)) == want
    assert lib.fiveeq_step_f64(None, 8, 8, None, 4, 0, None, None, None, None, None, None, 0, None, None) \
        == _capi.E_INVALID
    with pytest.raises(_capi.FiveEqError) as ei:
        _capi.check(lib, _call_step(lib, good, n=0))
    assert ei.value.code == _capi.E_INVALID


def test_run_and_plan_validation():
    lib = _capi.load()
    m = prm.make_model(prm.default_params("multigas"))
    p = ctypes.c_void_p(0x1000)
    args = lambda tb, te, rows=0: (ctypes.byref(m), 8, 8, p, 10, tb, te, p, p, p, p, None, None, rows, None)  # noqa: E731
    assert lib.fiveeq_run_f64(*args(3, 2), None) == _capi.E_INVALID
    assert lib.fiveeq_run_f32(*args(0, 11), None) == _capi.E_INVALID
    assert lib.fiveeq_run_fused_f64(*args(-1, 2), None) == _capi.E_INVALID
    assert lib.fiveeq_run_f64(*args(5, 5), None) == _capi.OK            # empty range: nothing launched
    assert lib.fiveeq_run_fused_f32(*args(5, 5), None) == _capi.OK
    assert lib.fiveeq_plan_create_f64(*args(0, 11), None) == _capi.E_INVALID
    assert lib.fiveeq_run_f64(*args(0, 5, -1), None) == _capi.E_INVALID        # n_rows < 0
    assert lib.fiveeq_plan_launch(None, None) == _capi.E_INVALID
    assert lib.fiveeq_plan_destroy(None) == _capi.E_INVALID
    assert lib.fiveeq_hfc_conc_f64(0, 0, 1, p, p, p, None) == _capi.E_INVALID
    assert lib.fiveeq_hfc_conc_f64(4, 4, 0, None, None, None, None) == _capi.OK  # no time points: nothing to do
    assert lib.fiveeq_hfc_conc_f64(4, 4, 3, None, p, p, None) == _capi.E_INVALID


def test_engine_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fiveeqscm_amd.engine import EnsembleEngine
    from fiveeqscm_amd.emissions import rcp_like_emissions
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        EnsembleEngine(prm.default_params("co2"), 16, rcp_like_emissions(10, 1))


def test_drive_table_output_rows():
    import numpy as np
    from fiveeqscm_amd.emissions import make_drive
    E = np.ones((6, 2))
    assert make_drive(E)[:, 7].tolist() == [0, 1, 2, 3, 4, 5]
    assert make_drive(E, output_steps=[4, 1, 4])[:, 7].tolist() == [-1, 0, -1, -1, 1, -1]
    assert make_drive(E, output_steps=[])[:, 7].tolist() == [-1] * 6
    with pytest.raises(ValueError):
        make_drive(E, output_steps=[6])
    assert make_drive(E, dt=0.5)[:, 3].tolist() == [0, 0.5, 1.0, 1.5, 2.0, 2.5]


def test_make_model_packs_the_parameter_dict():
    p = prm.default_params("multigas")
    m = prm.make_model(p, dt=0.5)
    assert m.n_gas == 3 and m.dt == 0.5 and m.iirf_max == 97.0
    assert [m.gas[g].n_pools for g in range(3)] == [4, 1, 1] == prm.pools_of(p)
    assert list(m.gas[0].tau) == p["tau"][0] and m.gas[1].tau[0] == 9.15
    assert abs(m.gas[0].g1 - 11.4137) < 1e-3 and abs(m.gas[0].g0 - 0.010184) < 1e-5
    with pytest.raises(ValueError):
        prm.make_model(dict(p, a=[[0.0, 1.0, 0, 0]] * 3))          # active pools must lead


def test_latin_hypercube_is_stratified_and_seeded():
    import numpy as np
    u = prm.latin_hypercube(1000, 5)
    assert u.shape == (5, 1000) and np.all((u > 0) & (u < 1))
    for k in range(5):
        assert np.array_equal(np.sort(np.floor(u[k] * 1000).astype(int)), np.arange(1000))
    assert np.array_equal(u, prm.latin_hypercube(1000, 5))
    s = prm.sample_ensemble(prm.default_params("multigas"), 5000)
    assert np.all(s["ECS"] >= 1.1 * s["TCR"] - 1e-12) and np.all(s["q"] > 0)
    assert s["r0"].shape == (3, 5000) and s["q"].shape == (2, 5000)
    r0c = np.asarray(prm.default_params("multigas")["r0"])[:, None]
    assert np.all(s["r0"] >= 0.8 * r0c - 1e-12) and np.all(s["r0"] <= 1.2 * r0c + 1e-12)


def test_new_entry_points_validate_on_the_host():
    """K-steps and the small-ensemble form.  Bad arguments return on the host (fake pointers are never dereferenced,
    nothing is launched)."""
    lib = _capi.load()
    m = prm.make_model(prm.default_params("multigas"))
    p = ctypes.c_void_p(0x1000)
    common = (ctypes.byref(m), 8, 8, p, 4, 0, 4, p, p, p, p, None, None, 0, None)
    assert lib.fiveeq_run_ksteps_f64(*common, 0, None) == _capi.E_INVALID and b"k_steps" in lib.fiveeq_last_error()
    assert lib.fiveeq_run_ksteps_f32(*common, -3, None) == _capi.E_INVALID
    # the compensated fp32 form (ABI v11): the arguments of run_ksteps + run_fused_bins in one entry point; bin_ring = NULL: no ring
    comp = lambda k, lo=0.0, hi=1.0, nb=1, ring=None, rows=0, t0=0, t1=4: lib.fiveeq_run_fused_comp_f32(   # noqa: E731
        ctypes.byref(m), 8, 8, p, 4, t0, t1, p, p, p, p, None, None, 0, None, k, lo, hi, nb, ring, rows, None)
    assert comp(0) == _capi.E_INVALID and b"k_steps" in lib.fiveeq_last_error()
    assert comp(4, t0=2, t1=2) == _capi.OK                                    # an empty range: validated, nothing launched
    assert comp(4, 1.0, 1.0, 8, p, 4) == _capi.E_INVALID and b"lo < hi" in lib.fiveeq_last_error()
    assert comp(4, 0.0, 1.0, 5000, p, 4) == _capi.E_INVALID and b"n_bins" in lib.fiveeq_last_error()
    assert comp(4, 0.0, 1.0, 8, p, 0) == _capi.E_INVALID and b"ring_rows" in lib.fiveeq_last_error()
    assert comp(4, 0.0, 1.0, 8, ctypes.c_void_p(0x1001), 4) == _capi.E_INVALID and b"aligned" in lib.fiveeq_last_error()
    assert comp(4, 0.0, 1.0, 8, p, 4, t0=3, t1=3) == _capi.OK
    assert lib.fiveeq_run_fused_comp_f32(None, 8, 8, p, 4, 0, 4, p, p, p, p, None, None, 0, None, 4, 0.0, 1.0, 1, None, 0, None) == _capi.E_INVALID
    small_comp = lambda t0, t1, Rp=p: lib.fiveeq_run_small_comp_f32(ctypes.byref(m), 8, 8, p, 4, t0, t1, p, p, Rp, p, None, None, 0, None, None)   # noqa: E731
    assert small_comp(2, 2) == _capi.OK and small_comp(0, 4, None) == _capi.E_INVALID and small_comp(3, 2) == _capi.E_INVALID
    # the small-ensemble kernels: 4 lanes per member for a lone 4-pool gas, 8 for 4 + 1 + 1, 1 for every other compiled layout
    lanes = lambda *pl: lib.fiveeq_small_lanes(len(pl), (ctypes.c_int32 * len(pl))(*pl))   # noqa: E731
    assert [lanes(4), lanes(1), lanes(2), lanes(3), lanes(4, 1, 1), lanes(1, 1), lanes(4, 4, 4), lanes(5), lanes(2, 3)] == [4, 1, 1, 1, 8, 1, 1, 0, 0]
    small = lambda model, n_lanes, t0=0, t1=4: lib.fiveeq_run_small_f64(ctypes.byref(model), 8, 8, p, 4, t0, t1, p, p, p, p, None,   # noqa: E731
                                                                       None, 0, None, n_lanes, None)
    assert small(m, 4) == _capi.E_INVALID and b"1 or 8" in lib.fiveeq_last_error()              # 4 + 1 + 1: one lane or an octet
    assert small(m, 0, 2, 2) == _capi.OK and small(m, 1, 2, 2) == _capi.OK and small(m, 8, 2, 2) == _capi.OK
    with_stats = lambda n_lanes: lib.fiveeq_run_small_f64(ctypes.byref(m), 8, 8, p, 4, 2, 2, p, p, p, p, None, None, 0, p, n_lanes, None)   # noqa: E731
    assert with_stats(8) == _capi.E_INVALID and b"statistics" in lib.fiveeq_last_error()       # the octet form writes no records
    assert with_stats(0) == _capi.OK and with_stats(1) == _capi.OK                              # "widest" then means one lane
    co2 = prm.make_model(prm.default_params("co2"))
    assert small(co2, 2) == _capi.E_INVALID and b"lanes_per_member" in lib.fiveeq_last_error()
    assert small(co2, -1) == _capi.E_INVALID
    assert small(co2, 4, 3, 2) == _capi.E_INVALID                      # the shared checks: step range
    for n_lanes in (0, 1, 4):
        assert small(co2, n_lanes, 2, 2) == _capi.OK                   # empty span: nothing to launch
    assert lib.fiveeq_run_small_f32(ctypes.byref(co2), 8, 8, p, 4, 2, 2, p, p, p, p, None, None, 0, None, 0, None) == _capi.OK
    one_pool = dict(prm.default_params("co2"))
    bad = lib.fiveeq_run_small_f64(ctypes.byref(co2), 8, 8, None, 4, 0, 4, p, p, p, p, None, None, 0, None, 0, None)
    assert bad == _capi.E_INVALID and b"NULL" in lib.fiveeq_last_error() and one_pool


def test_an_experiment_build_is_refused_unless_asked_for(monkeypatch):
    """A variant compiled with experiment knobs carries the product's source hash; fiveeq_build_flags() tells it apart and
    load() refuses it (advisor, round 4)."""
    lib = _capi.load()

    class Variant:
        def __getattr__(self, name):
            if name == "fiveeq_build_flags":
                return lambda: b" FIVEEQ_SMALL_BLOCK=64"
            return getattr(lib, name)

    monkeypatch.setattr(ctypes, "CDLL", lambda path: Variant())
    monkeypatch.delenv("FIVEEQ_ALLOW_STALE_LIB", raising=False)
    with pytest.raises(ImportError, match="experiment build"):
        _capi.load(_capi.LIB_PATH)
    monkeypatch.setenv("FIVEEQ_ALLOW_STALE_LIB", "1")
    assert _capi.build_flags(_capi.load(_capi.LIB_PATH)) == "FIVEEQ_SMALL_BLOCK=64"


def test_concurrent_calls_and_the_packing_switch():
    """include/fiveeq.h, CONVENTIONS: safe to call from several threads; the only process-wide state is the fp32 packing
    switch, an atomic.  Eight threads hammer the validation paths of fiveeq_step_f32 / fiveeq_run_small_f32 (every call fails
    or is an empty span: nothing is launched) while a ninth flips the switch; every thread must see ITS OWN error text
    (thread-local) and the switch must end where the last writer left it.  tools/sanitize_host.sh runs this under the
    ASan + UBSan host build."""
    import threading
    lib = _capi.load()
    m = prm.make_model(prm.default_params("multigas"))
    p = ctypes.c_void_p(0x1000)
    errors, stop = [], threading.Event()

    def worker(k):
        try:
            for i in range(3000):
                if k % 2:
                    rc = lib.fiveeq_step_f32(ctypes.byref(m), 8, 8, p, 4, 4 + k, p, p, p, p, None, None, 0, None, None)
                    want = f"t={4 + k} outside".encode()
                else:
                    rc = lib.fiveeq_run_f32(ctypes.byref(m), 8, 4 - (k % 3), p, 4, 0, 4, p, p, p, p, None, None, 0, None, None)
                    want = f"ld={4 - (k % 3)} <".encode()
                if rc != _capi.E_INVALID or want not in lib.fiveeq_last_error():
                    errors.append((k, i, rc, lib.fiveeq_last_error()))
                    return
                if lib.fiveeq_run_f32(ctypes.byref(m), 8, 8, p, 4, 2, 2, p, p, p, p, None, None, 0, None, None) != _capi.OK:
                    errors.append((k, i, "empty span"))
                    return
        except Exception as exc:  # noqa: BLE001
            errors.append((k, repr(exc)))

    def flipper():
        v = 0
        while not stop.is_set():
            lib.fiveeq_set_f32_packing(v)
            lib.fiveeq_set_row_policy(v)                       # the other process-wide word (ABI v10)
            v ^= 1

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    flip = threading.Thread(target=flipper)
    flip.start()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    stop.set()
    flip.join()
    assert not errors, errors[:3]
    assert lib.fiveeq_set_f32_packing(1) in (0, 1) and lib.fiveeq_set_f32_packing(1) == 1
    assert lib.fiveeq_set_row_policy(2) in (0, 1) and lib.fiveeq_set_row_policy(2) == 2


def test_row_policy_rule_and_switch():
    """include/fiveeq.h, "CACHE POLICY OF THE PER-STEP KERNEL'S ROWS": the rule as documented (a launch takes the streamed
    form when its own rows fill the 256 MiB Infinity Cache and the ensemble is at least two caches), on the shapes the engine
    produces — whole ensembles, the halves of a two-stream split, the chunks of the chunk-major schedule — and the switch."""
    torch = pytest.importorskip("torch")
    from fiveeqscm_amd.engine import EnsembleEngine
    lib = _capi.load()
    multigas, co2 = (ctypes.c_int32 * 3)(4, 1, 1), (ctypes.c_int32 * 1)(4)
    q = lambda pools, G, n, ld, w: lib.fiveeq_rows_streamed(G, pools, n, ld, w)   # noqa: E731
    assert lib.fiveeq_set_row_policy(2) in (0, 1, 2)                              # FIVEEQ_ROWS_AUTO
    assert [q(multigas, 3, n, n, 8) for n in (1_000_000, 2_000_000, 3_000_000, 4_000_000, 8_000_000)] == [0, 0, 0, 1, 1]
    assert q(multigas, 3, 2_000_000, 4_000_000, 8) == 1 and q(multigas, 3, 1_000_000, 4_000_000, 8) == 0   # halves / quarters of 4M
    assert q(multigas, 3, 12_500_000, 12_500_000, 4) == 1 and q(co2, 1, 1_000_000, 1_000_000, 8) == 0
    for n_total, dtype, w in ((8_000_000, torch.float64, 8), (12_500_000, torch.float32, 4), (100_000_000, torch.float32, 4)):
        chunk = EnsembleEngine.auto_chunk(n_total, 6, 3, dtype)
        assert chunk and q(multigas, 3, chunk, n_total, w) == 0, (n_total, chunk)   # chunk-major launches stay cache-resident
    assert lib.fiveeq_set_row_policy(1) == 2 and q(co2, 1, 64, 64, 8) == 1        # forced: streamed
    assert lib.fiveeq_set_row_policy(0) == 1 and q(multigas, 3, 8_000_000, 8_000_000, 8) == 0
    assert lib.fiveeq_set_row_policy(5) == _capi.E_INVALID and b"row policy 5" in lib.fiveeq_last_error()
    assert lib.fiveeq_set_row_policy(2) == 0                                       # the rejected value changed nothing
    assert lib.fiveeq_rows_streamed(4, multigas, 10, 10, 8) == _capi.E_INVALID and lib.fiveeq_rows_streamed(3, multigas, 10, 5, 8) == _capi.E_INVALID
    assert lib.fiveeq_rows_streamed(3, multigas, 10, 10, 2) == _capi.E_INVALID


def test_abi_v5_host_side_guards():
    """K-step spans are clamped to the step range before any loop arithmetic, and Latin-hypercube designs stop at 2^28
    members, where stratum + jitter is still an exact fp64 sum."""
    import numpy as np
    lib = _capi.load()
    m = prm.make_model(prm.default_params("multigas"))
    p = ctypes.c_void_p(0x1000)
    empty = (ctypes.byref(m), 8, 8, p, 4, 2, 2, p, p, p, p, None, None, 0, None)       # t_begin == t_end: nothing to launch
    assert lib.fiveeq_run_ksteps_f64(*empty, 2**31 - 1, None) == _capi.OK
    assert lib.fiveeq_run_ksteps_f32(*empty, 1, None) == _capi.OK
    out = np.empty((1, 4))
    host = lambda n_total: lib.fiveeq_lhs_rows_host_f64(1, n_total, 0, 4, 0, 1, 4, out.ctypes.data_as(ctypes.c_void_p))  # noqa: E731
    assert host(1 << 28) == _capi.OK and np.all((out > 0) & (out < 1))
    assert host((1 << 28) + 1) == _capi.E_INVALID and b"2^28" in lib.fiveeq_last_error()
    assert lib.fiveeq_lhs_rows_f64(1, 1 << 40, 0, 4, 0, 1, 4, p, None) == _capi.E_INVALID
    with pytest.raises(ValueError, match="2\\^28"):
        prm.lhs_rows((1 << 28) + 1, [0], 0, 4)
    # at the cap the twins still agree bit for bit and every u sits strictly inside its stratum
    n = 1 << 28
    u = prm.lhs_rows(n, [0, 1], n - 1000, n)
    got = np.empty_like(u)
    assert lib.fiveeq_lhs_rows_host_f64(prm.LHS_SEED, n, n - 1000, 1000, 0, 2, 1000, got.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.array_equal(u, got)
    scaled = u * n
    assert np.all(scaled - np.floor(scaled) > 0) and np.all(np.floor(scaled) < n)


def test_abi_v6_bin_ring_entry_points_validate_on_the_host():
    """fiveeq_run_fused_bins_* / fiveeq_hist_bins: bad arguments return on the host, nothing is launched."""
    lib = _capi.load()
    m = prm.make_model(prm.default_params("multigas"))
    p = ctypes.c_void_p(0x1000)
    common = (ctypes.byref(m), 8, 8, p, 4, 0, 4, p, p, p, p, None, None, 0, None)
    bins = lambda lo, hi, nb, ring, rows: lib.fiveeq_run_fused_bins_f64(*common, lo, hi, nb, ring, rows, None)   # noqa: E731
    assert bins(0.0, 1.0, 0, p, 4) == _capi.E_INVALID and b"n_bins" in lib.fiveeq_last_error()
    assert bins(0.0, 1.0, 4097, p, 4) == _capi.E_INVALID
    assert bins(1.0, 1.0, 16, p, 4) == _capi.E_INVALID and b"lo < hi" in lib.fiveeq_last_error()
    assert bins(0.0, float("nan"), 16, p, 4) == _capi.E_INVALID
    assert bins(0.0, 1.0, 16, None, 4) == _capi.E_INVALID and b"bin_ring" in lib.fiveeq_last_error()
    assert bins(0.0, 1.0, 16, p, 0) == _capi.E_INVALID and b"ring_rows" in lib.fiveeq_last_error()
    assert bins(0.0, 1.0, 16, ctypes.c_void_p(0x1001), 4) == _capi.E_INVALID and b"aligned" in lib.fiveeq_last_error()
    empty = (ctypes.byref(m), 8, 8, p, 4, 2, 2, p, p, p, p, None, None, 0, None)
    assert lib.fiveeq_run_fused_bins_f32(*empty, 0.0, 1.0, 16, p, 4, None) == _capi.OK           # empty span: nothing to do
    # the per-step form shares the checks
    assert lib.fiveeq_run_bins_f64(*common, 0.0, 1.0, 16, None, 4, None) == _capi.E_INVALID and b"bin_ring" in lib.fiveeq_last_error()
    assert lib.fiveeq_run_bins_f32(*common, 2.0, 1.0, 16, p, 4, None) == _capi.E_INVALID
    assert lib.fiveeq_run_bins_f32(*empty, 0.0, 1.0, 16, p, 4, None) == _capi.OK
    count = lambda rows, n, ld, b, nb, h: lib.fiveeq_hist_bins(rows, n, ld, b, nb, h, None)       # noqa: E731
    assert count(-1, 8, 8, p, 16, p) == _capi.E_INVALID and count(2, 0, 8, p, 16, p) == _capi.E_INVALID
    assert count(2, 9, 8, p, 16, p) == _capi.E_INVALID and count(2, 8, 8, p, 0, p) == _capi.E_INVALID
    assert count(2, 8, 8, None, 16, p) == _capi.E_INVALID and count(2, 8, 8, p, 16, None) == _capi.E_INVALID
    assert count(70000, 8, 8, p, 16, p) == _capi.E_INVALID and count(0, 8, 8, None, 16, None) == _capi.OK


def test_engine_module_validates_its_environment():
    """FIVEEQ_HBM_STREAM_BYTES_PER_S / FIVEEQ_LAUNCH_BOUNDARY_S: garbage, zero or negative values keep the defaults with a
    warning instead of breaking the import or dividing by zero later; FIVEEQ_HIST_PASS_STREAM accepts 'side' / 'same' only."""
    import subprocess
    import sys
    code = ("import warnings; warnings.simplefilter('error'); import fiveeqscm_amd.engine as e; "
            "print(e.HBM_STREAM_BYTES_PER_S, e.LAUNCH_BOUNDARY_S)")
    env = {k: v for k, v in os.environ.items() if not k.startswith("FIVEEQ_")}
    ok = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                        env=dict(env, FIVEEQ_HBM_STREAM_BYTES_PER_S="5e12", FIVEEQ_LAUNCH_BOUNDARY_S="3e-6"))
    assert ok.returncode == 0 and ok.stdout.split() == ["5000000000000.0", "3e-06"], ok.stderr
    for bad in ("", "abc", "0", "-2e-6", "nan", "inf"):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                             env=dict(env, FIVEEQ_LAUNCH_BOUNDARY_S=bad))
        assert out.returncode != 0 and "not a positive number" in out.stderr, (bad, out.stderr)      # the warning, made an error
        soft = subprocess.run([sys.executable, "-W", "ignore", "-c", code.replace("warnings.simplefilter('error'); ", "")],
                              capture_output=True, text=True, cwd=ROOT, env=dict(env, FIVEEQ_LAUNCH_BOUNDARY_S=bad))
        assert soft.returncode == 0 and soft.stdout.split()[1] == "2e-06", (bad, soft.stderr)
    from fiveeqscm_amd import engine
    with pytest.raises(ValueError, match="FIVEEQ_HIST_PASS_STREAM"):
        os.environ["FIVEEQ_HIST_PASS_STREAM"] = "sidee"
        try:
            engine._env_choice("FIVEEQ_HIST_PASS_STREAM", "side", ("side", "same"))
        finally:
            del os.environ["FIVEEQ_HIST_PASS_STREAM"]


def test_the_ctypes_stub_in_integration_md_binds_the_built_library():
    """INTEGRATION.md shows the binding a maintainer of the reference would add.  Execute that very code block against the
    built library: its struct mirror, ABI number and argument list must be the library's (the document cannot drift)."""
    with open(os.path.join(ROOT, "INTEGRATION.md")) as fh:
        text = fh.read()
    block = re.search(r"```python\n(# U_FaIR/_fiveeq_hip\.py.*?)```", text, re.S).group(1)
    assert 'ctypes.CDLL("libfiveeq_hip.so")' in block
    code = block.replace('ctypes.CDLL("libfiveeq_hip.so")', f"ctypes.CDLL({_capi.LIB_PATH!r})")
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)                  # its own asserts: ABI version, sizeof(model), build flags
    assert ctypes.sizeof(ns["Model"]) == ctypes.sizeof(_capi.Model) == 448 and callable(ns["run"])
    assert [f[0] for f in ns["Gas"]._fields_] == [f[0] for f in _capi.Gas._fields_]
    assert len(ns["lib"].fiveeq_run_f64.argtypes) == len(_capi.SIGNATURES["fiveeq_run_f64"][1])
