"""ctypes binding of the C ABI in include/fiveeq.h (libfiveeq_hip.so).

There is NO fallback: if the HIP library is missing or does not export the
expected symbols, `load()` raises.  The ensemble engine cannot run without it.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FIVEEQ_LIB_PATH selects another build of the same library (e.g. the host-sanitizer build of tools/sanitize_host.sh)
LIB_PATH = os.environ.get("FIVEEQ_LIB_PATH") or os.path.join(_HERE, "csrc", "libfiveeq_hip.so")

ABI_VERSION = 11
MAX_GAS = 3
MAX_POOLS = 4
N_BOX = 2
DRIVE_STRIDE = 8
LHS_MAX_TOTAL = 1 << 28

OK = 0
E_INVALID = -1
E_UNSUPPORTED = -2
E_HIP = -3


class FiveEqError(RuntimeError):
    """A C-ABI call returned a negative code; `.code` holds it."""

    def __init__(self, code, message):
        super().__init__(f"fiveeq error {code}: {message}")
        self.code = code


class Gas(ctypes.Structure):
    """struct fiveeq_gas (include/fiveeq.h)."""
    _fields_ = [
        ("a", ctypes.c_double * MAX_POOLS),
        ("tau", ctypes.c_double * MAX_POOLS),
        ("g0", ctypes.c_double),
        ("g1", ctypes.c_double),
        ("ra", ctypes.c_double),
        ("C0", ctypes.c_double),
        ("emis2conc", ctypes.c_double),
        ("f", ctypes.c_double * 3),
        ("n_pools", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


class Model(ctypes.Structure):
    """struct fiveeq_model (include/fiveeq.h)."""
    _fields_ = [
        ("gas", Gas * MAX_GAS),
        ("d", ctypes.c_double * N_BOX),
        ("iirf_max", ctypes.c_double),
        ("dt", ctypes.c_double),
        ("n_gas", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
    ]


_p = ctypes.c_void_p
_i64 = ctypes.c_int64
_i32 = ctypes.c_int32
_mp = ctypes.POINTER(Model)

# name -> (restype, argtypes); every symbol include/fiveeq.h declares
# (model, n, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, T_stats, stream|plan_out)
_RUN_ARGS = [_mp, _i64, _i64, _p, _i32, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _p, _p]
_STEP_ARGS = [_mp, _i64, _i64, _p, _i32, _i32, _p, _p, _p, _p, _p, _p, _i32, _p, _p]
SIGNATURES = {
    "fiveeq_abi_version": (ctypes.c_int, []),
    "fiveeq_last_error": (ctypes.c_char_p, []),
    "fiveeq_source_hash": (ctypes.c_char_p, []),
    "fiveeq_build_flags": (ctypes.c_char_p, []),
    "fiveeq_sizeof_model": (ctypes.c_int64, []),
    "fiveeq_layout_supported": (ctypes.c_int, [_i32, ctypes.POINTER(_i32)]),
    "fiveeq_stats_waves": (ctypes.c_int64, [_i64]),
    "fiveeq_step_f64": (ctypes.c_int, _STEP_ARGS),
    "fiveeq_step_f32": (ctypes.c_int, _STEP_ARGS),
    "fiveeq_run_f64": (ctypes.c_int, _RUN_ARGS),
    "fiveeq_run_f32": (ctypes.c_int, _RUN_ARGS),
    "fiveeq_run_fused_f64": (ctypes.c_int, _RUN_ARGS),
    "fiveeq_run_fused_f32": (ctypes.c_int, _RUN_ARGS),
    "fiveeq_run_fused_bins_f64": (ctypes.c_int, _RUN_ARGS[:-1] + [ctypes.c_double, ctypes.c_double, _i32, _p, _i32, _p]),
    "fiveeq_run_fused_bins_f32": (ctypes.c_int, _RUN_ARGS[:-1] + [ctypes.c_double, ctypes.c_double, _i32, _p, _i32, _p]),
    "fiveeq_hist_bins": (ctypes.c_int, [_i32, _i64, _i64, _p, _i32, _p, _p]),
    "fiveeq_run_bins_f64": (ctypes.c_int, _RUN_ARGS[:-1] + [ctypes.c_double, ctypes.c_double, _i32, _p, _i32, _p]),
    "fiveeq_run_bins_f32": (ctypes.c_int, _RUN_ARGS[:-1] + [ctypes.c_double, ctypes.c_double, _i32, _p, _i32, _p]),
    "fiveeq_plan_create_f64": (ctypes.c_int, _RUN_ARGS[:-1] + [ctypes.POINTER(_p)]),
    "fiveeq_plan_create_f32": (ctypes.c_int, _RUN_ARGS[:-1] + [ctypes.POINTER(_p)]),
    "fiveeq_run_inverse_f64": (ctypes.c_int, _RUN_ARGS[:11] + [_p] + _RUN_ARGS[11:]),
    "fiveeq_run_inverse_f32": (ctypes.c_int, _RUN_ARGS[:11] + [_p] + _RUN_ARGS[11:]),
    "fiveeq_run_ksteps_f64": (ctypes.c_int, _RUN_ARGS[:-1] + [_i32, _p]),
    "fiveeq_run_ksteps_f32": (ctypes.c_int, _RUN_ARGS[:-1] + [_i32, _p]),
    "fiveeq_run_small_f64": (ctypes.c_int, _RUN_ARGS[:-1] + [_i32, _p]),
    "fiveeq_run_small_f32": (ctypes.c_int, _RUN_ARGS[:-1] + [_i32, _p]),
    "fiveeq_run_fused_comp_f32": (ctypes.c_int, _RUN_ARGS[:-1] + [_i32, ctypes.c_double, ctypes.c_double, _i32, _p, _i32, _p]),
    "fiveeq_run_small_comp_f32": (ctypes.c_int, _RUN_ARGS),
    "fiveeq_small_lanes": (_i32, [_i32, ctypes.POINTER(_i32)]),
    "fiveeq_set_f32_packing": (ctypes.c_int, [ctypes.c_int]),
    "fiveeq_set_row_policy": (ctypes.c_int, [_i32]),
    "fiveeq_rows_streamed": (ctypes.c_int, [_i32, ctypes.POINTER(_i32), _i64, _i64, _i32]),
    "fiveeq_lhs_rows_f64": (ctypes.c_int, [ctypes.c_uint64, _i64, _i64, _i64, _i32, _i32, _i64, _p, _p]),
    "fiveeq_lhs_rows_host_f64": (ctypes.c_int, [ctypes.c_uint64, _i64, _i64, _i64, _i32, _i32, _i64, _p]),
    "fiveeq_plan_launch": (ctypes.c_int, [_p, _p]),
    "fiveeq_plan_destroy": (ctypes.c_int, [_p]),
    "fiveeq_hfc_conc_f64": (ctypes.c_int, [_i64, _i64, _i32, _p, _p, _p, _p]),
    "fiveeq_hist_rows_f64": (ctypes.c_int, [_i32, _i64, _i64, _p, ctypes.c_double, ctypes.c_double, _i32, _p, _p]),
    "fiveeq_hist_rows_f32": (ctypes.c_int, [_i32, _i64, _i64, _p, ctypes.c_double, ctypes.c_double, _i32, _p, _p]),
    "fiveeq_row_moments_chunks": (ctypes.c_int64, [_i32, _i64]),
    "fiveeq_row_moments_f64": (ctypes.c_int, [_i32, _i64, _i64, _p, _p, _p, _p]),
    "fiveeq_row_moments_f32": (ctypes.c_int, [_i32, _i64, _i64, _p, _p, _p, _p]),
    "fiveeq_hist_rows_ranged_f64": (ctypes.c_int, [_i32, _i64, _i64, _p, _p, _i32, _p, _p]),
    "fiveeq_hist_rows_ranged_f32": (ctypes.c_int, [_i32, _i64, _i64, _p, _p, _i32, _p, _p]),
    "fiveeq_select_bins_f64": (ctypes.c_int, [_i32, _i64, _i64, _p, _p, _i32, _p, _p, _i64, _p, _p]),
    "fiveeq_select_bins_f32": (ctypes.c_int, [_i32, _i64, _i64, _p, _p, _i32, _p, _p, _i64, _p, _p]),
    "fiveeq_select_pick_f64": (ctypes.c_int, [_i32, _i32, _i64, _p, _p, _i32, _p, _p, _p]),
    "fiveeq_select_pick_f32": (ctypes.c_int, [_i32, _i32, _i64, _p, _p, _i32, _p, _p, _p]),
    "fiveeq_stream_copy_f64": (ctypes.c_int, [_i64, _p, _p, _p]),
    "fiveeq_stream_copy_wide_f64": (ctypes.c_int, [_i64, _p, _p, _p]),
    "fiveeq_stream_copy_nt_f64": (ctypes.c_int, [_i64, _p, _p, _p]),
    "fiveeq_busy": (ctypes.c_int, [_i64, _p, _p]),
    "fiveeq_math_probe_f64": (ctypes.c_int, [_i32, _i64, _p, _p, _p]),
    "fiveeq_math_probe_f32": (ctypes.c_int, [_i32, _i64, _p, _p, _p]),
}

_lib = None
SOURCES = (os.path.join(_HERE, "csrc", "fiveeq_capi.hip"), os.path.join(_HERE, "csrc", "fiveeq_device.hpp"),
           os.path.join(os.path.dirname(_HERE), "include", "fiveeq.h"))


def source_hash():
    """sha256 (hex) of the library's three sources as they lie in this tree, concatenated in SOURCES order — what
    csrc/Makefile stamps into the library; None when the tree carries no sources (an installed binary)."""
    import hashlib
    if not all(os.path.exists(p) for p in SOURCES):
        return None
    h = hashlib.sha256()
    for p in SOURCES:
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def load(path=None):
    """Load libfiveeq_hip.so and bind every entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    lib_path = path or LIB_PATH
    if not os.path.exists(lib_path):
        raise ImportError(
            f"{lib_path} not found: the HIP extension is not built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or `make -C fiveeqscm_amd/csrc`). "
            "There is no CPU fallback for the ensemble engine.")
    # torch (if present) must be imported first so that its bundled libamdhip64.so.7 is the
    # HIP runtime this library binds to by soname: one runtime per process, so torch tensor
    # pointers and torch streams are valid in our launches.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(lib_path)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise ImportError(f"{lib_path} does not export {name}") from exc
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.fiveeq_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{lib_path}: ABI version {got}, expected {ABI_VERSION}")
    if lib.fiveeq_sizeof_model() != ctypes.sizeof(Model):
        raise ImportError(f"{lib_path}: sizeof(fiveeq_model)={lib.fiveeq_sizeof_model()} but the ctypes "
                          f"mirror is {ctypes.sizeof(Model)} bytes")
    # The library must have been compiled from the sources lying next to this file: a prebuilt .so that travelled to
    # another box, or survived a source edit, is refused instead of tested.  (FIVEEQ_ALLOW_STALE_LIB=1: experiment
    # variants built from patched sources, tools/ only.)
    want, got_hash = source_hash(), (lib.fiveeq_source_hash() or b"").decode()
    if want is not None and got_hash != want and os.environ.get("FIVEEQ_ALLOW_STALE_LIB") != "1":
        raise ImportError(f"{lib_path} was built from other sources (library {got_hash[:16]}, tree {want[:16]}): "
                          "run `make -C fiveeqscm_amd/csrc`")
    # ... and it must be the PRODUCT build: a variant compiled with experiment knobs (csrc/Makefile EXTRA=-D...) carries the same
    # source hash, and only fiveeq_build_flags() tells it apart
    flags = (lib.fiveeq_build_flags() or b"").decode().strip()
    if flags and os.environ.get("FIVEEQ_ALLOW_STALE_LIB") != "1":
        raise ImportError(f"{lib_path} is an experiment build ({flags}); set FIVEEQ_ALLOW_STALE_LIB=1 to load it (tools/ only)")
    if path is None:
        _lib = lib
    return lib


def build_flags(lib=None):
    """Experiment knobs the loaded library was compiled with ('' for the product build)."""
    return ((lib or load()).fiveeq_build_flags() or b"").decode().strip()


def check(lib, rc):
    if rc != OK:
        msg = lib.fiveeq_last_error()
        raise FiveEqError(rc, msg.decode("utf-8", "replace") if msg else "")
