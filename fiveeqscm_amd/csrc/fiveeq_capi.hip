// fiveeq_capi.hip — the C ABI declared in include/fiveeq.h: host-side validation,
// model preparation and kernel dispatch.  gfx950 only; build with
//   hipcc --offload-arch=gfx950 -O3 -fPIC -shared -I include fiveeq_capi.hip -o libfiveeq_hip.so
#include "fiveeq.h"
#include "fiveeq_device.hpp"

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>

#ifndef FIVEEQ_FUSED_DYN_LDS
#define FIVEEQ_FUSED_DYN_LDS 0     // experiment knob: unused dynamic LDS per fused workgroup, to cap occupancy
#endif

namespace {

using namespace fiveeq;

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(FIVEEQ_E_HIP, "%s failed: %s (%d)", #expr, hipGetErrorString(e_), (int)e_); \
    } while (0)

// ---- pool layouts with a compiled kernel --------------------------------------------------
// X(P0, P1, P2): pools of gas 0, 1, 2 (0 = gas absent).  CO2-like gases carry 4 pools,
// single-lifetime gases (CH4, N2O, HFCs) carry 1.
#define FIVEEQ_LAYOUTS(X) \
    X(1, 0, 0) X(2, 0, 0) X(3, 0, 0) X(4, 0, 0) \
    X(1, 1, 0) X(4, 1, 0) X(4, 4, 0)            \
    X(1, 1, 1) X(4, 1, 1) X(4, 4, 1) X(4, 4, 4)

int layout_code(const fiveeq_model* m) {
    int p[3] = {0, 0, 0};
    for (int g = 0; g < m->n_gas; ++g) p[g] = m->gas[g].n_pools;
    return p[0] * 100 + p[1] * 10 + p[2];
}

bool layout_ok(int code) {
    switch (code) {
#define X(a, b, c) case (a) * 100 + (b) * 10 + (c):
        FIVEEQ_LAYOUTS(X)
#undef X
        return true;
        default:
            return false;
    }
}

// ---- validation (host; nothing reaches the GPU unless this passes) -------------------------
int check_model(const fiveeq_model* m) {
    if (!m) return fail(FIVEEQ_E_INVALID, "model is NULL");
    if (m->n_gas < 1 || m->n_gas > FIVEEQ_MAX_GAS)
        return fail(FIVEEQ_E_INVALID, "n_gas=%d outside 1..%d", m->n_gas, FIVEEQ_MAX_GAS);
    if (!(m->dt > 0.0) || !std::isfinite(m->dt)) return fail(FIVEEQ_E_INVALID, "dt=%g must be finite and > 0", m->dt);
    for (int j = 0; j < FIVEEQ_N_BOX; ++j)
        if (!(m->d[j] > 0.0) || !std::isfinite(m->d[j]))
            return fail(FIVEEQ_E_INVALID, "d[%d]=%g must be finite and > 0", j, m->d[j]);
    if (std::isnan(m->iirf_max)) return fail(FIVEEQ_E_INVALID, "iirf_max is NaN");
    for (int g = 0; g < m->n_gas; ++g) {
        const fiveeq_gas& gs = m->gas[g];
        if (gs.n_pools < 1 || gs.n_pools > FIVEEQ_MAX_POOLS)
            return fail(FIVEEQ_E_INVALID, "gas %d: n_pools=%d outside 1..%d", g, gs.n_pools, FIVEEQ_MAX_POOLS);
        for (int i = 0; i < gs.n_pools; ++i)
            if (!(gs.tau[i] > 0.0) || !std::isfinite(gs.tau[i]) || !std::isfinite(gs.a[i]))
                return fail(FIVEEQ_E_INVALID, "gas %d pool %d: a=%g tau=%g invalid", g, i, gs.a[i], gs.tau[i]);
        if (!(gs.g1 != 0.0) || !std::isfinite(gs.g1) || !std::isfinite(gs.g0))
            return fail(FIVEEQ_E_INVALID, "gas %d: g0=%g g1=%g invalid", g, gs.g0, gs.g1);
        if (!(gs.C0 > 0.0) || !std::isfinite(gs.C0)) return fail(FIVEEQ_E_INVALID, "gas %d: C0=%g must be > 0", g, gs.C0);
        if (!(gs.emis2conc > 0.0) || !std::isfinite(gs.emis2conc))
            return fail(FIVEEQ_E_INVALID, "gas %d: emis2conc=%g must be > 0", g, gs.emis2conc);
    }
    if (!layout_ok(layout_code(m)))
        return fail(FIVEEQ_E_UNSUPPORTED, "pool layout %03d has no compiled kernel", layout_code(m));
    return FIVEEQ_OK;
}

int check_run(const fiveeq_model* m, int64_t n, int64_t ld, const void* drive, int32_t n_steps, int32_t t_begin,
              int32_t t_end, const void* r, const void* q, const void* R, const void* S) {
    if (int rc = check_model(m)) return rc;
    if (n < 1) return fail(FIVEEQ_E_INVALID, "n_members=%lld must be >= 1", (long long)n);
    if (ld < n) return fail(FIVEEQ_E_INVALID, "ld=%lld < n_members=%lld", (long long)ld, (long long)n);
    if (n_steps < 1) return fail(FIVEEQ_E_INVALID, "n_steps=%d must be >= 1", n_steps);
    if (t_begin < 0 || t_end > n_steps || t_begin > t_end)
        return fail(FIVEEQ_E_INVALID, "step range [%d,%d) outside [0,%d)", t_begin, t_end, n_steps);
    if (!drive || !r || !q || !R || !S)
        return fail(FIVEEQ_E_INVALID, "NULL device pointer (drive=%p r=%p q=%p R=%p S=%p)", drive, r, q, R, S);
    return FIVEEQ_OK;
}

// ---- model -> kernel-precision constants ---------------------------------------------------
template <typename T>
KModel<T> make_kmodel(const fiveeq_model* m) {
    KModel<T> km;
    std::memset(&km, 0, sizeof km);
    for (int g = 0; g < m->n_gas; ++g) {
        const fiveeq_gas& gs = m->gas[g];
        KGas<T>& kg = km.gas[g];
        for (int i = 0; i < gs.n_pools; ++i) {
            kg.ndt_over_tau[i] = (T)(-m->dt / gs.tau[i]);
            kg.atc[i] = (T)(gs.a[i] * gs.tau[i] * gs.emis2conc);
        }
        kg.g0 = (T)gs.g0;
        kg.inv_g1 = (T)(1.0 / gs.g1);
        kg.ra = (T)gs.ra;
        kg.inv_c = (T)(1.0 / gs.emis2conc);
        kg.C0 = (T)gs.C0;
        kg.inv_C0 = (T)(1.0 / gs.C0);
        kg.sqrtC0 = (T)std::sqrt(gs.C0);
        kg.f1 = (T)gs.f[0];
        kg.f2 = (T)gs.f[1];
        kg.f3 = (T)gs.f[2];
    }
    for (int j = 0; j < 2; ++j) km.em1_d[j] = (T)std::expm1(-m->dt / m->d[j]);
    km.iirf_max = (T)m->iirf_max;
    km.dt = (T)m->dt;
    return km;
}

// One member per lane, one 256-thread workgroup per 256 members (3907 workgroups at 1M members,
// >> 256 CUs); workgroup b owns the same members in every launch.
int64_t member_blocks(int64_t n) { return (n + FIVEEQ_BLOCK - 1) / FIVEEQ_BLOCK; }

template <typename T>
struct RunArgs {
    KModel<T> km;
    int code;
    int n_gas;
    int64_t n, ld;
    const T* drive;
    const T *r, *q;
    T *R, *S, *C_traj, *T_traj;
    int n_rows;
    int n_steps;
    double* stats;
    int packing;                 // the fp32 packing switch as this call found it
    bool stream_rows;            // the per-step launches of this call take the STREAMED (non-temporal) row form
};

// the bin-index ring of the streamed histograms (fused_kernel<..., BINS = true>)
struct BinRing {
    unsigned short* ring = nullptr;
    int ring_rows = 0;
    double lo = 0.0, inv_w = 0.0;
    int n_bins = 0;
};

// ---- packed fp32 lanes: two members per lane (fiveeq_device.hpp, "Lane value types") ----------------------------
// The fp32 entry points run the packed kernels whenever the rows allow 8-byte accesses: even row stride, every row
// pointer 8-byte aligned, at least two members.  Otherwise (odd ld, a sub-range starting at an odd member) the
// one-member-per-lane kernels run; both give the same bits.  fiveeq_set_f32_packing(0) forces the scalar kernels (A/B
// measurements, and the tests that compare the two).
// Process-wide and changeable at any time from any thread: a relaxed atomic, read ONCE per C-ABI call (make_args), so that
// every launch of one call takes the same kernel shape.
std::atomic<int> g_f32_packing{1};

// ---- cache policy of the per-step kernel's state and parameter rows (fiveeq_device.hpp, step_kernel<..., NT>) --------------
// STREAMED (non-temporal) pays exactly when the rows of this launch cannot be in the Infinity Cache at the next step:
//   * the launch's own rows fill it: n members x (SP + 2 + 3G + 2) words >= the cache — false for the chunks of a chunk-major
//     schedule (the engine sizes them to fit), true for an unchunked multi-million-member launch; and
//   * the ensemble they are a part of (row length ld: the halves of a two-stream split share the cache) is at least twice
//     the cache — between one and two cache sizes the default policy still hits often enough to win (2M fp64 members, 304 MB:
//     +13 % streamed; 4M: -9 %; profiles/r05/step_row_policy_ab.txt).
// fiveeq_set_row_policy overrides the rule process-wide (A/B measurements, the bit-identity tests); like the packing switch
// it is a relaxed atomic read ONCE per C-ABI call.
constexpr int64_t INFINITY_CACHE_BYTES = (int64_t)256 << 20;       // MI355X (MI355X_MICROARCH.md)
std::atomic<int> g_row_policy{FIVEEQ_ROWS_AUTO};

bool rows_streamed(int policy, int n_gas, int sum_pools, int64_t n, int64_t ld, int word) {
    if (policy != FIVEEQ_ROWS_AUTO) return policy == FIVEEQ_ROWS_STREAMED;
    const int64_t per_member = (int64_t)word * (sum_pools + 2 + 3 * n_gas + 2);
    return n * per_member >= INFINITY_CACHE_BYTES && ld * per_member >= 2 * INFINITY_CACHE_BYTES;
}

template <typename T>
struct LaneOf {
    using Packed = T;                              // fp64 has no packed VALU forms: one member per lane
    static bool can_pack(const RunArgs<T>&) { return false; }
};
template <>
struct LaneOf<float> {
    using Packed = float2v;
    static bool can_pack(const RunArgs<float>& a) {
        if (!a.packing || a.n < 2 || (a.ld & 1)) return false;
        const uintptr_t bits = (uintptr_t)a.r | (uintptr_t)a.q | (uintptr_t)a.R | (uintptr_t)a.S | (uintptr_t)a.C_traj |
                               (uintptr_t)a.T_traj;
        return (bits & 7u) == 0;
    }
};

template <typename T, bool BINS = false>
int launch_step(const RunArgs<T>& a, int t, hipStream_t st, const BinRing& br = BinRing()) {
    using P = typename LaneOf<T>::Packed;
    const bool packed = LaneOf<T>::can_pack(a) && (!BINS || (((uintptr_t)br.ring) & 3) == 0);
    const int64_t per_block = (int64_t)FIVEEQ_STEP_BLOCK * (packed ? 2 : 1);
    const int64_t blocks = (a.n + per_block - 1) / per_block;
    if (blocks > 0x7fffffffLL) return fail(FIVEEQ_E_INVALID, "n_members too large for one launch");
    const dim3 grid((unsigned)blocks), block(FIVEEQ_STEP_BLOCK);
    switch (a.code) {
#define FIVEEQ_STEP_LAUNCH(V, p0, p1, p2, NT)                                                                                  \
    hipLaunchKernelGGL((step_kernel<V, p0, p1, p2, BINS, NT>), grid, block, 0, st, a.km, a.drive, a.n_steps, t, a.n, a.ld, a.r, \
                       a.q, a.R, a.S, a.C_traj, a.T_traj, a.n_rows, a.stats, br.ring, br.ring_rows, br.lo, br.inv_w, br.n_bins)
#define X(p0, p1, p2)                                                                             \
    case (p0) * 100 + (p1) * 10 + (p2):                                                           \
        if constexpr (!BINS) {            /* the streamed row form: plain per-step launches only */ \
            if (a.stream_rows) {                                                                  \
                if (packed) FIVEEQ_STEP_LAUNCH(P, p0, p1, p2, true);                              \
                else FIVEEQ_STEP_LAUNCH(T, p0, p1, p2, true);                                     \
                break;                                                                            \
            }                                                                                     \
        }                                                                                         \
        if (packed) FIVEEQ_STEP_LAUNCH(P, p0, p1, p2, false);                                     \
        else FIVEEQ_STEP_LAUNCH(T, p0, p1, p2, false);                                            \
        break;
        FIVEEQ_LAYOUTS(X)
#undef X
#undef FIVEEQ_STEP_LAUNCH
        default:
            return fail(FIVEEQ_E_UNSUPPORTED, "pool layout %03d has no compiled kernel", a.code);
    }
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}

template <typename T, bool INV, bool BINS = false, bool COMP = false>
int launch_fused(const RunArgs<T>& a, int t_begin, int t_end, T* cumE, hipStream_t st, const BinRing& br = BinRing()) {
    using P = typename LaneOf<T>::Packed;
    constexpr bool HAS_PACKED = !INV && !std::is_same<P, T>::value;     // the inverse form has no packed instantiation
    // packed lanes store two 2-byte bin indices as one 4-byte word: the ring rows must be 4-byte aligned too
    const bool packed = HAS_PACKED && LaneOf<T>::can_pack(a) && (!BINS || (((uintptr_t)br.ring) & 3) == 0);
    const int64_t blocks = member_blocks(packed ? (a.n + 1) / 2 : a.n);
    if (blocks > 0x7fffffffLL) return fail(FIVEEQ_E_INVALID, "n_members too large for one launch");
    const dim3 grid((unsigned)blocks), block(FIVEEQ_BLOCK);
    switch (a.code) {
#define X(p0, p1, p2)                                                                                  \
    case (p0) * 100 + (p1) * 10 + (p2):                                                                \
        if constexpr (HAS_PACKED) {                                                                    \
            if (packed) {                                                                              \
                hipLaunchKernelGGL((fused_kernel<P, p0, p1, p2, false, BINS, COMP>), grid, block, FIVEEQ_FUSED_DYN_LDS, st, a.km,  \
                                   a.drive, a.n_steps, t_begin, t_end, a.n, a.ld, a.r, a.q, a.R, a.S, cumE, a.C_traj,        \
                                   a.T_traj, a.n_rows, a.stats, br.ring, br.ring_rows, br.lo, br.inv_w, br.n_bins);         \
                break;                                                                                 \
            }                                                                                          \
        }                                                                                              \
        hipLaunchKernelGGL((fused_kernel<T, p0, p1, p2, INV, BINS, COMP>), grid, block, FIVEEQ_FUSED_DYN_LDS, st, a.km, a.drive,   \
                           a.n_steps, t_begin, t_end, a.n, a.ld, a.r, a.q, a.R, a.S, cumE, a.C_traj, a.T_traj, a.n_rows,    \
                           a.stats, br.ring, br.ring_rows, br.lo, br.inv_w, br.n_bins);                \
        break;
        FIVEEQ_LAYOUTS(X)
#undef X
        default:
            return fail(FIVEEQ_E_UNSUPPORTED, "pool layout %03d has no compiled kernel", a.code);
    }
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}

template <typename T>
int make_args(RunArgs<T>& a, const fiveeq_model* m, int64_t n, int64_t ld, const T* drive, int32_t n_steps,
              int32_t t_begin, int32_t t_end, const T* r, const T* q, T* R, T* S, T* C_traj, T* T_traj, int n_rows, double* stats) {
    if (int rc = check_run(m, n, ld, drive, n_steps, t_begin, t_end, r, q, R, S)) return rc;
    if (n_rows < 0) return fail(FIVEEQ_E_INVALID, "n_rows=%d must be >= 0", n_rows);
    a.km = make_kmodel<T>(m);
    a.code = layout_code(m);
    a.n_gas = m->n_gas;
    a.n = n;
    a.ld = ld;
    a.drive = drive;
    a.r = r;
    a.q = q;
    a.R = R;
    a.S = S;
    a.C_traj = C_traj;
    a.T_traj = T_traj;
    a.n_rows = n_rows;
    a.n_steps = n_steps;
    a.stats = stats;
    a.packing = g_f32_packing.load(std::memory_order_relaxed);
    int sum_pools = 0;
    for (int g = 0; g < m->n_gas; ++g) sum_pools += m->gas[g].n_pools;
    a.stream_rows = rows_streamed(g_row_policy.load(std::memory_order_relaxed), m->n_gas, sum_pools, n, ld, (int)sizeof(T));
    return FIVEEQ_OK;
}

template <typename T>
int run_steps(const fiveeq_model* m, int64_t n, int64_t ld, const T* drive, int32_t n_steps, int32_t t_begin,
              int32_t t_end, const T* r, const T* q, T* R, T* S, T* C_traj, T* T_traj, int n_rows, double* stats, void* stream) {
    RunArgs<T> a;
    if (int rc = make_args(a, m, n, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, stats)) return rc;
    for (int t = t_begin; t < t_end; ++t)
        if (int rc = launch_step(a, t, (hipStream_t)stream)) return rc;
    return FIVEEQ_OK;
}

template <typename T>
int run_fused(const fiveeq_model* m, int64_t n, int64_t ld, const T* drive, int32_t n_steps, int32_t t_begin,
              int32_t t_end, const T* r, const T* q, T* R, T* S, T* C_traj, T* T_traj, int n_rows, double* stats, void* stream) {
    RunArgs<T> a;
    if (int rc = make_args(a, m, n, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, stats)) return rc;
    if (t_begin == t_end) return FIVEEQ_OK;
    return launch_fused<T, false>(a, t_begin, t_end, nullptr, (hipStream_t)stream);
}

template <typename T, bool FUSED>
int run_bins(const fiveeq_model* m, int64_t n, int64_t ld, const T* drive, int32_t n_steps, int32_t t_begin,
             int32_t t_end, const T* r, const T* q, T* R, T* S, T* C_traj, T* T_traj, int n_rows, double* stats,
             double lo, double hi, int32_t n_bins, uint16_t* bin_ring, int32_t ring_rows, void* stream) {
    RunArgs<T> a;
    if (int rc = make_args(a, m, n, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, stats)) return rc;
    if (n_bins < 1 || n_bins > HIST_MAX_BINS) return fail(FIVEEQ_E_INVALID, "n_bins=%d outside 1..%d", n_bins, HIST_MAX_BINS);
    if (!(hi > lo) || !std::isfinite(lo) || !std::isfinite(hi)) return fail(FIVEEQ_E_INVALID, "need finite lo < hi");
    if (!bin_ring) return fail(FIVEEQ_E_INVALID, "bin_ring is NULL");
    if (ring_rows < 1) return fail(FIVEEQ_E_INVALID, "ring_rows=%d must be >= 1", ring_rows);
    if (((uintptr_t)bin_ring) & 1) return fail(FIVEEQ_E_INVALID, "bin_ring must be 2-byte aligned");
    if (t_begin == t_end) return FIVEEQ_OK;
    BinRing br;
    br.ring = bin_ring;
    br.ring_rows = ring_rows;
    br.lo = lo;
    br.inv_w = (double)n_bins / (hi - lo);
    br.n_bins = n_bins;
    if (FUSED) return launch_fused<T, false, true>(a, t_begin, t_end, nullptr, (hipStream_t)stream, br);
    for (int t = t_begin; t < t_end; ++t)                 // the per-step form: one launch per timestep
        if (int rc = launch_step<T, true>(a, t, (hipStream_t)stream, br)) return rc;
    return FIVEEQ_OK;
}

template <typename T>
int run_inverse(const fiveeq_model* m, int64_t n, int64_t ld, const T* drive, int32_t n_steps, int32_t t_begin,
                int32_t t_end, const T* r, const T* q, T* R, T* S, T* cumE, T* E_traj, T* T_traj, int n_rows,
                double* stats, void* stream) {
    RunArgs<T> a;
    if (int rc = make_args(a, m, n, ld, drive, n_steps, t_begin, t_end, r, q, R, S, E_traj, T_traj, n_rows, stats)) return rc;
    if (!cumE) return fail(FIVEEQ_E_INVALID, "cumE is NULL");
    if (t_begin == t_end) return FIVEEQ_OK;
    return launch_fused<T, true>(a, t_begin, t_end, cumE, (hipStream_t)stream);
}

// ---- K steps per launch: the fused kernel over consecutive spans of k_steps ---------------------
template <typename T>
int run_ksteps(const fiveeq_model* m, int64_t n, int64_t ld, const T* drive, int32_t n_steps, int32_t t_begin,
               int32_t t_end, const T* r, const T* q, T* R, T* S, T* C_traj, T* T_traj, int n_rows, double* stats,
               int32_t k_steps, void* stream) {
    RunArgs<T> a;
    if (int rc = make_args(a, m, n, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, stats)) return rc;
    if (k_steps < 1) return fail(FIVEEQ_E_INVALID, "k_steps=%d must be >= 1", k_steps);
    if (t_begin == t_end) return FIVEEQ_OK;
    if (k_steps > t_end - t_begin) k_steps = t_end - t_begin;      // also keeps t + k_steps inside int32
    for (int t = t_begin; t < t_end; t += k_steps)
        if (int rc = launch_fused<T, false>(a, t, t + k_steps < t_end ? t + k_steps : t_end, nullptr, (hipStream_t)stream))
            return rc;
    return FIVEEQ_OK;
}

// ---- the compensated fp32 form: the fused kernel <.., COMP = true> over spans of k_steps, with or without the bin ring ----
int run_fused_comp(const fiveeq_model* m, int64_t n, int64_t ld, const float* drive, int32_t n_steps, int32_t t_begin,
                   int32_t t_end, const float* r, const float* q, float* R, float* S, float* C_traj, float* T_traj, int n_rows,
                   double* stats, int32_t k_steps, double lo, double hi, int32_t n_bins, uint16_t* bin_ring, int32_t ring_rows,
                   void* stream) {
    RunArgs<float> a;
    if (int rc = make_args(a, m, n, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, stats)) return rc;
    if (k_steps < 1) return fail(FIVEEQ_E_INVALID, "k_steps=%d must be >= 1", k_steps);
    BinRing br;
    if (bin_ring) {
        if (n_bins < 1 || n_bins > HIST_MAX_BINS) return fail(FIVEEQ_E_INVALID, "n_bins=%d outside 1..%d", n_bins, HIST_MAX_BINS);
        if (!(hi > lo) || !std::isfinite(lo) || !std::isfinite(hi)) return fail(FIVEEQ_E_INVALID, "need finite lo < hi");
        if (ring_rows < 1) return fail(FIVEEQ_E_INVALID, "ring_rows=%d must be >= 1", ring_rows);
        if (((uintptr_t)bin_ring) & 1) return fail(FIVEEQ_E_INVALID, "bin_ring must be 2-byte aligned");
        br.ring = bin_ring;
        br.ring_rows = ring_rows;
        br.lo = lo;
        br.inv_w = (double)n_bins / (hi - lo);
        br.n_bins = n_bins;
    }
    if (t_begin == t_end) return FIVEEQ_OK;
    if (k_steps > t_end - t_begin) k_steps = t_end - t_begin;
    for (int t = t_begin; t < t_end; t += k_steps) {
        const int t1 = t + k_steps < t_end ? t + k_steps : t_end;
        const int rc = bin_ring ? launch_fused<float, false, true, true>(a, t, t1, nullptr, (hipStream_t)stream, br)
                                : launch_fused<float, false, false, true>(a, t, t1, nullptr, (hipStream_t)stream);
        if (rc) return rc;
    }
    return FIVEEQ_OK;
}

// ---- small ensembles: one member per quad of lanes (small_kernel), several gases one per lane (small_multi_kernel) -----
// lanes per member of the widest small-ensemble form compiled for a layout: 4 for a lone 4-pool gas (a quad), 8 for 4 + 1 + 1
// (an octet: small_octet_kernel), 1 for every other compiled layout, 0 = none
int small_lanes(int code) { return code == 400 ? 4 : (code == 411 ? 8 : (layout_ok(code) ? 1 : 0)); }

template <typename T>
int run_small(const fiveeq_model* m, int64_t n, int64_t ld, const T* drive, int32_t n_steps, int32_t t_begin,
              int32_t t_end, const T* r, const T* q, T* R, T* S, T* C_traj, T* T_traj, int n_rows, double* stats,
              int32_t lanes, void* stream) {
    RunArgs<T> a;
    if (int rc = make_args(a, m, n, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, stats)) return rc;
    const int widest = small_lanes(a.code);
    if (lanes == 0) lanes = (widest == 8 && a.stats != nullptr) ? 1 : widest;      // the octet form writes no statistics records
    if (lanes != 1 && lanes != widest)
        return fail(FIVEEQ_E_INVALID, "lanes_per_member=%d: pool layout %03d takes 1%s", lanes, a.code,
                    widest == 4 ? " or 4" : (widest == 8 ? " or 8" : ""));
    if (lanes == 8 && a.stats != nullptr)
        return fail(FIVEEQ_E_INVALID, "lanes_per_member=8 writes no per-wave statistics (T_stats must be NULL): use 1");
    if (t_begin == t_end) return FIVEEQ_OK;
    const int64_t per_block = FIVEEQ_SMALL_BLOCK / lanes;
    const int64_t blocks = (a.n + per_block - 1) / per_block;
    if (blocks > 0x7fffffffLL) return fail(FIVEEQ_E_INVALID, "n_members too large for one launch");
    const dim3 grid((unsigned)blocks), block(FIVEEQ_SMALL_BLOCK);
    hipStream_t st = (hipStream_t)stream;
#define FIVEEQ_SMALL_ARGS st, a.km, a.drive, a.n_steps, t_begin, t_end, a.n, a.ld, a.r, a.q, a.R, a.S, a.C_traj, a.T_traj, a.n_rows, a.stats
    const bool st_on = a.stats != nullptr;
#define FIVEEQ_SMALL1(p0, lpm)                                                                                     \
    if (st_on) hipLaunchKernelGGL((small_kernel<T, p0, lpm, true>), grid, block, 0, FIVEEQ_SMALL_ARGS);            \
    else hipLaunchKernelGGL((small_kernel<T, p0, lpm, false>), grid, block, 0, FIVEEQ_SMALL_ARGS);                 \
    break;
    switch (a.code * 10 + lanes) {
        case 1001: FIVEEQ_SMALL1(1, 1)
        case 2001: FIVEEQ_SMALL1(2, 1)
        case 3001: FIVEEQ_SMALL1(3, 1)
        case 4001: FIVEEQ_SMALL1(4, 1)
        case 4004: FIVEEQ_SMALL1(4, 4)
        case 4118:
            hipLaunchKernelGGL((small_octet_kernel<T>), grid, block, 0, st, a.km, a.drive, a.n_steps, t_begin, t_end, a.n, a.ld, a.r, a.q,
                               a.R, a.S, a.C_traj, a.T_traj, a.n_rows);
            break;
#define X(p0, p1, p2)                                                                                              \
    case ((p0) * 100 + (p1) * 10 + (p2)) * 10 + 1:                                                                 \
        if constexpr ((p1) > 0) {                                                                                  \
            if (st_on) hipLaunchKernelGGL((small_multi_kernel<T, p0, p1, p2, true>), grid, block, 0, FIVEEQ_SMALL_ARGS);  \
            else hipLaunchKernelGGL((small_multi_kernel<T, p0, p1, p2, false>), grid, block, 0, FIVEEQ_SMALL_ARGS);       \
        }                                                                                                          \
        break;
        X(1, 1, 0) X(4, 1, 0) X(4, 4, 0) X(1, 1, 1) X(4, 1, 1) X(4, 4, 1) X(4, 4, 4)
#undef X
        default: return fail(FIVEEQ_E_UNSUPPORTED, "pool layout %03d has no compiled kernel", a.code);
    }
#undef FIVEEQ_SMALL1
#undef FIVEEQ_SMALL_ARGS
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}

// ---- the compensated fp32 form on the small-ensemble kernel: one member per lane, every layout ----
int run_small_comp(const fiveeq_model* m, int64_t n, int64_t ld, const float* drive, int32_t n_steps, int32_t t_begin,
                   int32_t t_end, const float* r, const float* q, float* R, float* S, float* C_traj, float* T_traj, int n_rows,
                   double* stats, void* stream) {
    RunArgs<float> a;
    if (int rc = make_args(a, m, n, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, stats)) return rc;
    if (t_begin == t_end) return FIVEEQ_OK;
    const int64_t blocks = (a.n + FIVEEQ_SMALL_BLOCK - 1) / FIVEEQ_SMALL_BLOCK;
    if (blocks > 0x7fffffffLL) return fail(FIVEEQ_E_INVALID, "n_members too large for one launch");
    const dim3 grid((unsigned)blocks), block(FIVEEQ_SMALL_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    const bool st_on = a.stats != nullptr;
    switch (a.code) {
#define X(p0, p1, p2)                                                                                                          \
    case (p0) * 100 + (p1) * 10 + (p2):                                                                                        \
        if (st_on)                                                                                                             \
            hipLaunchKernelGGL((small_multi_kernel<float, p0, p1, p2, true, true>), grid, block, 0, st, a.km, a.drive, a.n_steps,  \
                               t_begin, t_end, a.n, a.ld, a.r, a.q, a.R, a.S, a.C_traj, a.T_traj, a.n_rows, a.stats);          \
        else                                                                                                                   \
            hipLaunchKernelGGL((small_multi_kernel<float, p0, p1, p2, false, true>), grid, block, 0, st, a.km, a.drive, a.n_steps, \
                               t_begin, t_end, a.n, a.ld, a.r, a.q, a.R, a.S, a.C_traj, a.T_traj, a.n_rows, a.stats);          \
        break;
        FIVEEQ_LAYOUTS(X)
#undef X
        default: return fail(FIVEEQ_E_UNSUPPORTED, "pool layout %03d has no compiled kernel", a.code);
    }
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}

// ---- plans: the per-step launch sequence captured into a hipGraph --------------------------
struct Plan {
    uint32_t magic;
    hipGraph_t graph;
    hipGraphExec_t exec;
};
constexpr uint32_t PLAN_MAGIC = 0x35455146u;  // "FQE5"

template <typename T>
int plan_create(const fiveeq_model* m, int64_t n, int64_t ld, const T* drive, int32_t n_steps, int32_t t_begin,
                int32_t t_end, const T* r, const T* q, T* R, T* S, T* C_traj, T* T_traj, int n_rows, double* stats, void** plan_out) {
    if (!plan_out) return fail(FIVEEQ_E_INVALID, "plan_out is NULL");
    *plan_out = nullptr;
    RunArgs<T> a;
    if (int rc = make_args(a, m, n, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, stats)) return rc;
    if (t_begin == t_end) return fail(FIVEEQ_E_INVALID, "empty step range for a plan");
    hipStream_t cap = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(cap);
        return fail(FIVEEQ_E_HIP, "hipStreamBeginCapture failed: %s", hipGetErrorString(e));
    }
    int rc = FIVEEQ_OK;
    for (int t = t_begin; t < t_end && rc == FIVEEQ_OK; ++t) rc = launch_step(a, t, cap);
    e = hipStreamEndCapture(cap, &graph);
    (void)hipStreamDestroy(cap);
    if (rc != FIVEEQ_OK) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
    }
    if (e != hipSuccess) return fail(FIVEEQ_E_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(graph);
        return fail(FIVEEQ_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    }
    Plan* p = new (std::nothrow) Plan{PLAN_MAGIC, graph, exec};
    if (!p) {
        (void)hipGraphExecDestroy(exec);
        (void)hipGraphDestroy(graph);
        return fail(FIVEEQ_E_INVALID, "out of host memory");
    }
    *plan_out = p;
    return FIVEEQ_OK;
}

}  // namespace

// =============================================================================================
extern "C" {

int fiveeq_abi_version(void) { return FIVEEQ_ABI_VERSION; }
#ifndef FIVEEQ_SOURCE_HASH
#define FIVEEQ_SOURCE_HASH "unstamped"      // built outside csrc/Makefile: the Python binding refuses such a library
#endif
const char* fiveeq_source_hash(void) { return FIVEEQ_SOURCE_HASH; }
#define FIVEEQ_STR2(x) #x
#define FIVEEQ_STR(x) FIVEEQ_STR2(x)
const char* fiveeq_build_flags(void) {
    return ""
#ifdef FIVEEQ_STEP_WAVES
           " FIVEEQ_STEP_WAVES=" FIVEEQ_STR(FIVEEQ_STEP_WAVES)
#endif
#if FIVEEQ_FUSED_DYN_LDS != 0
           " FIVEEQ_FUSED_DYN_LDS=" FIVEEQ_STR(FIVEEQ_FUSED_DYN_LDS)
#endif
#if FIVEEQ_BLOCK != 256
           " FIVEEQ_BLOCK=" FIVEEQ_STR(FIVEEQ_BLOCK)
#endif
#if FIVEEQ_SMALL_BLOCK != 256
           " FIVEEQ_SMALL_BLOCK=" FIVEEQ_STR(FIVEEQ_SMALL_BLOCK)
#endif
#if FIVEEQ_STEP_BLOCK != 64
           " FIVEEQ_STEP_BLOCK=" FIVEEQ_STR(FIVEEQ_STEP_BLOCK)
#endif
#if FIVEEQ_FUSED_CHUNK != 125
           " FIVEEQ_FUSED_CHUNK=" FIVEEQ_STR(FIVEEQ_FUSED_CHUNK)
#endif
        ;
}
const char* fiveeq_last_error(void) { return g_err; }
int64_t fiveeq_sizeof_model(void) { return (int64_t)sizeof(fiveeq_model); }
int64_t fiveeq_stats_waves(int64_t n_members) { return n_members < 1 ? 0 : (n_members + 63) / 64; }

int fiveeq_layout_supported(int32_t n_gas, const int32_t* n_pools) {
    if (!n_pools || n_gas < 1 || n_gas > FIVEEQ_MAX_GAS) return 0;
    int p[3] = {0, 0, 0};
    for (int g = 0; g < n_gas; ++g) {
        if (n_pools[g] < 1 || n_pools[g] > FIVEEQ_MAX_POOLS) return 0;
        p[g] = n_pools[g];
    }
    return layout_ok(p[0] * 100 + p[1] * 10 + p[2]) ? 1 : 0;
}

int fiveeq_step_f64(const fiveeq_model* model, int64_t n_members, int64_t ld, const double* drive, int32_t n_steps,
                    int32_t t, const double* r, const double* q, double* R, double* S, double* C_traj, double* T_traj,
                    int32_t n_rows, double* T_stats, void* stream) {
    if (t < 0 || t >= n_steps) return fail(FIVEEQ_E_INVALID, "t=%d outside [0,%d)", t, n_steps);
    return run_steps<double>(model, n_members, ld, drive, n_steps, t, t + 1, r, q, R, S, C_traj, T_traj, n_rows, T_stats, stream);
}
int fiveeq_run_f64(const fiveeq_model* model, int64_t n_members, int64_t ld, const double* drive, int32_t n_steps,
                   int32_t t_begin, int32_t t_end, const double* r, const double* q, double* R, double* S, double* C_traj,
                   double* T_traj, int32_t n_rows, double* T_stats, void* stream) {
    return run_steps<double>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, T_stats,
                        stream);
}
int fiveeq_run_fused_f64(const fiveeq_model* model, int64_t n_members, int64_t ld, const double* drive,
                         int32_t n_steps, int32_t t_begin, int32_t t_end, const double* r, const double* q, double* R,
                         double* S, double* C_traj, double* T_traj, int32_t n_rows, double* T_stats, void* stream) {
    return run_fused<double>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, T_stats,
                        stream);
}
int fiveeq_plan_create_f64(const fiveeq_model* model, int64_t n_members, int64_t ld, const double* drive,
                           int32_t n_steps, int32_t t_begin, int32_t t_end, const double* r, const double* q, double* R,
                           double* S, double* C_traj, double* T_traj, int32_t n_rows, double* T_stats, void** plan_out) {
    return plan_create<double>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, T_stats,
                          plan_out);
}
int fiveeq_step_f32(const fiveeq_model* model, int64_t n_members, int64_t ld, const float* drive, int32_t n_steps,
                    int32_t t, const float* r, const float* q, float* R, float* S, float* C_traj, float* T_traj,
                    int32_t n_rows, double* T_stats, void* stream) {
    if (t < 0 || t >= n_steps) return fail(FIVEEQ_E_INVALID, "t=%d outside [0,%d)", t, n_steps);
    return run_steps<float>(model, n_members, ld, drive, n_steps, t, t + 1, r, q, R, S, C_traj, T_traj, n_rows, T_stats, stream);
}
int fiveeq_run_f32(const fiveeq_model* model, int64_t n_members, int64_t ld, const float* drive, int32_t n_steps,
                   int32_t t_begin, int32_t t_end, const float* r, const float* q, float* R, float* S, float* C_traj,
                   float* T_traj, int32_t n_rows, double* T_stats, void* stream) {
    return run_steps<float>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, T_stats,
                        stream);
}
int fiveeq_run_fused_f32(const fiveeq_model* model, int64_t n_members, int64_t ld, const float* drive,
                         int32_t n_steps, int32_t t_begin, int32_t t_end, const float* r, const float* q, float* R,
                         float* S, float* C_traj, float* T_traj, int32_t n_rows, double* T_stats, void* stream) {
    return run_fused<float>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, T_stats,
                        stream);
}
int fiveeq_run_fused_bins_f64(const fiveeq_model* model, int64_t n_members, int64_t ld, const double* drive,
                              int32_t n_steps, int32_t t_begin, int32_t t_end, const double* r, const double* q, double* R,
                              double* S, double* C_traj, double* T_traj, int32_t n_rows, double* T_stats, double hist_lo,
                              double hist_hi, int32_t n_bins, uint16_t* bin_ring, int32_t ring_rows, void* stream) {
    return run_bins<double, true>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows,
                                  T_stats, hist_lo, hist_hi, n_bins, bin_ring, ring_rows, stream);
}
int fiveeq_run_fused_bins_f32(const fiveeq_model* model, int64_t n_members, int64_t ld, const float* drive,
                              int32_t n_steps, int32_t t_begin, int32_t t_end, const float* r, const float* q, float* R,
                              float* S, float* C_traj, float* T_traj, int32_t n_rows, double* T_stats, double hist_lo,
                              double hist_hi, int32_t n_bins, uint16_t* bin_ring, int32_t ring_rows, void* stream) {
    return run_bins<float, true>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows,
                                 T_stats, hist_lo, hist_hi, n_bins, bin_ring, ring_rows, stream);
}
int fiveeq_run_bins_f64(const fiveeq_model* model, int64_t n_members, int64_t ld, const double* drive, int32_t n_steps,
                        int32_t t_begin, int32_t t_end, const double* r, const double* q, double* R, double* S,
                        double* C_traj, double* T_traj, int32_t n_rows, double* T_stats, double hist_lo, double hist_hi,
                        int32_t n_bins, uint16_t* bin_ring, int32_t ring_rows, void* stream) {
    return run_bins<double, false>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows,
                                   T_stats, hist_lo, hist_hi, n_bins, bin_ring, ring_rows, stream);
}
int fiveeq_run_bins_f32(const fiveeq_model* model, int64_t n_members, int64_t ld, const float* drive, int32_t n_steps,
                        int32_t t_begin, int32_t t_end, const float* r, const float* q, float* R, float* S,
                        float* C_traj, float* T_traj, int32_t n_rows, double* T_stats, double hist_lo, double hist_hi,
                        int32_t n_bins, uint16_t* bin_ring, int32_t ring_rows, void* stream) {
    return run_bins<float, false>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows,
                                  T_stats, hist_lo, hist_hi, n_bins, bin_ring, ring_rows, stream);
}
int fiveeq_plan_create_f32(const fiveeq_model* model, int64_t n_members, int64_t ld, const float* drive,
                           int32_t n_steps, int32_t t_begin, int32_t t_end, const float* r, const float* q, float* R,
                           float* S, float* C_traj, float* T_traj, int32_t n_rows, double* T_stats, void** plan_out) {
    return plan_create<float>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, T_stats,
                          plan_out);
}

int fiveeq_run_inverse_f64(const fiveeq_model* model, int64_t n_members, int64_t ld, const double* drive,
                           int32_t n_steps, int32_t t_begin, int32_t t_end, const double* r, const double* q,
                           double* R, double* S, double* cumE, double* E_traj, double* T_traj, int32_t n_rows,
                           double* T_stats, void* stream) {
    return run_inverse<double>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, cumE, E_traj, T_traj,
                               n_rows, T_stats, stream);
}
int fiveeq_run_inverse_f32(const fiveeq_model* model, int64_t n_members, int64_t ld, const float* drive,
                           int32_t n_steps, int32_t t_begin, int32_t t_end, const float* r, const float* q, float* R,
                           float* S, float* cumE, float* E_traj, float* T_traj, int32_t n_rows, double* T_stats,
                           void* stream) {
    return run_inverse<float>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, cumE, E_traj, T_traj,
                              n_rows, T_stats, stream);
}

int fiveeq_run_ksteps_f64(const fiveeq_model* model, int64_t n_members, int64_t ld, const double* drive,
                          int32_t n_steps, int32_t t_begin, int32_t t_end, const double* r, const double* q, double* R,
                          double* S, double* C_traj, double* T_traj, int32_t n_rows, double* T_stats, int32_t k_steps,
                          void* stream) {
    return run_ksteps<double>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows,
                              T_stats, k_steps, stream);
}
int fiveeq_run_ksteps_f32(const fiveeq_model* model, int64_t n_members, int64_t ld, const float* drive,
                          int32_t n_steps, int32_t t_begin, int32_t t_end, const float* r, const float* q, float* R,
                          float* S, float* C_traj, float* T_traj, int32_t n_rows, double* T_stats, int32_t k_steps,
                          void* stream) {
    return run_ksteps<float>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows,
                             T_stats, k_steps, stream);
}
int fiveeq_run_fused_comp_f32(const fiveeq_model* model, int64_t n_members, int64_t ld, const float* drive, int32_t n_steps,
                              int32_t t_begin, int32_t t_end, const float* r, const float* q, float* R, float* S, float* C_traj,
                              float* T_traj, int32_t n_rows, double* T_stats, int32_t k_steps, double lo, double hi,
                              int32_t n_bins, uint16_t* bin_ring, int32_t ring_rows, void* stream) {
    return run_fused_comp(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, T_stats,
                          k_steps, lo, hi, n_bins, bin_ring, ring_rows, stream);
}
int fiveeq_run_small_f64(const fiveeq_model* model, int64_t n_members, int64_t ld, const double* drive,
                         int32_t n_steps, int32_t t_begin, int32_t t_end, const double* r, const double* q, double* R,
                         double* S, double* C_traj, double* T_traj, int32_t n_rows, double* T_stats, int32_t lanes_per_member,
                         void* stream) {
    return run_small<double>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, T_stats,
                             lanes_per_member, stream);
}
int fiveeq_run_small_f32(const fiveeq_model* model, int64_t n_members, int64_t ld, const float* drive,
                         int32_t n_steps, int32_t t_begin, int32_t t_end, const float* r, const float* q, float* R,
                         float* S, float* C_traj, float* T_traj, int32_t n_rows, double* T_stats, int32_t lanes_per_member,
                         void* stream) {
    return run_small<float>(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, T_stats,
                            lanes_per_member, stream);
}
int fiveeq_run_small_comp_f32(const fiveeq_model* model, int64_t n_members, int64_t ld, const float* drive, int32_t n_steps,
                              int32_t t_begin, int32_t t_end, const float* r, const float* q, float* R, float* S, float* C_traj,
                              float* T_traj, int32_t n_rows, double* T_stats, void* stream) {
    return run_small_comp(model, n_members, ld, drive, n_steps, t_begin, t_end, r, q, R, S, C_traj, T_traj, n_rows, T_stats, stream);
}
int32_t fiveeq_small_lanes(int32_t n_gas, const int32_t* n_pools) {
    if (!fiveeq_layout_supported(n_gas, n_pools)) return 0;
    int p[3] = {0, 0, 0};
    for (int g = 0; g < n_gas; ++g) p[g] = n_pools[g];
    return small_lanes(p[0] * 100 + p[1] * 10 + p[2]);
}
int fiveeq_set_f32_packing(int on) { return g_f32_packing.exchange(on ? 1 : 0, std::memory_order_relaxed); }

int fiveeq_set_row_policy(int32_t policy) {
    if (policy != FIVEEQ_ROWS_AUTO && policy != FIVEEQ_ROWS_CACHED && policy != FIVEEQ_ROWS_STREAMED)
        return fail(FIVEEQ_E_INVALID, "row policy %d: want FIVEEQ_ROWS_CACHED (0), _STREAMED (1) or _AUTO (2)", policy);
    return g_row_policy.exchange(policy, std::memory_order_relaxed);
}

int fiveeq_rows_streamed(int32_t n_gas, const int32_t* n_pools, int64_t n_members, int64_t ld, int32_t word_bytes) {
    if (n_gas < 1 || n_gas > FIVEEQ_MAX_GAS || !n_pools || n_members < 0 || ld < n_members || (word_bytes != 4 && word_bytes != 8))
        return fail(FIVEEQ_E_INVALID, "fiveeq_rows_streamed: bad shape");
    int sum_pools = 0;
    for (int g = 0; g < n_gas; ++g) sum_pools += n_pools[g];
    return rows_streamed(g_row_policy.load(std::memory_order_relaxed), n_gas, sum_pools, n_members, ld, word_bytes) ? 1 : 0;
}

static int lhs_check(int64_t n_total, int64_t m0, int64_t n_members, int32_t dim0, int32_t n_dim, int64_t ld) {
    // 2^28: stratum (28 bits) + the 24-bit jitter placed mid-cell (25 fractional bits) is then an EXACT fp64 sum, so u lies
    // strictly inside its stratum; beyond that the sum would round and could touch the stratum edge
    if (n_total < 1 || n_total > FIVEEQ_LHS_MAX_TOTAL)
        return fail(FIVEEQ_E_INVALID, "n_total=%lld outside 1..2^28", (long long)n_total);
    if (m0 < 0 || n_members < 0 || m0 + n_members > n_total)
        return fail(FIVEEQ_E_INVALID, "members [%lld, %lld) outside [0, %lld)", (long long)m0, (long long)(m0 + n_members),
                    (long long)n_total);
    if (dim0 < 0 || n_dim < 0 || n_dim > 65535) return fail(FIVEEQ_E_INVALID, "dim0=%d n_dim=%d invalid", dim0, n_dim);
    if (ld < n_members) return fail(FIVEEQ_E_INVALID, "ld < n_members");
    return FIVEEQ_OK;
}
static int lhs_half_bits(int64_t n_total) {
    int bits = 2;                                            // even number of bits with 2^bits >= n_total
    while (bits < 62 && (1LL << bits) < n_total) bits += 2;
    return bits / 2;
}

int fiveeq_lhs_rows_host_f64(uint64_t seed, int64_t n_total, int64_t m0, int64_t n_members, int32_t dim0, int32_t n_dim,
                             int64_t ld, double* out) {
    if (int rc = lhs_check(n_total, m0, n_members, dim0, n_dim, ld)) return rc;
    if (n_members == 0 || n_dim == 0) return FIVEEQ_OK;
    if (!out) return fail(FIVEEQ_E_INVALID, "NULL pointer");
    const int half = lhs_half_bits(n_total);
    for (int k = 0; k < n_dim; ++k) {
        const uint64_t key = fiveeq::lhs_dim_key(seed, dim0 + k);
        for (int64_t i = 0; i < n_members; ++i) {
            const uint64_t m = (uint64_t)(m0 + i);
            const uint64_t stratum = fiveeq::lhs_permute(m, (uint64_t)n_total, half, key);
            const uint64_t jbits = fiveeq::lhs_mix64(m ^ (key * 0xff51afd7ed558ccdULL + 0xc4ceb9fe1a85ec53ULL)) >> 40;
            out[(int64_t)k * ld + i] = ((double)stratum + ((double)jbits + 0.5) * 0x1.0p-24) / (double)n_total;
        }
    }
    return FIVEEQ_OK;
}

int fiveeq_lhs_rows_f64(uint64_t seed, int64_t n_total, int64_t m0, int64_t n_members, int32_t dim0, int32_t n_dim,
                        int64_t ld, double* out, void* stream) {
    if (int rc = lhs_check(n_total, m0, n_members, dim0, n_dim, ld)) return rc;
    if (n_members == 0 || n_dim == 0) return FIVEEQ_OK;
    if (!out) return fail(FIVEEQ_E_INVALID, "NULL device pointer");
    const int64_t blocks = member_blocks(n_members);
    if (blocks > 0x7fffffffLL) return fail(FIVEEQ_E_INVALID, "n_members too large for one launch");
    hipLaunchKernelGGL(fiveeq::lhs_kernel, dim3((unsigned)blocks, (unsigned)n_dim), dim3(FIVEEQ_BLOCK), 0, (hipStream_t)stream,
                       seed, n_total, lhs_half_bits(n_total), m0, n_members, dim0, ld, out);
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}

int fiveeq_plan_launch(void* plan, void* stream) {
    Plan* p = static_cast<Plan*>(plan);
    if (!p || p->magic != PLAN_MAGIC) return fail(FIVEEQ_E_INVALID, "not a live fiveeq plan");
    HIP_TRY(hipGraphLaunch(p->exec, (hipStream_t)stream));
    return FIVEEQ_OK;
}

int fiveeq_plan_destroy(void* plan) {
    Plan* p = static_cast<Plan*>(plan);
    if (!p || p->magic != PLAN_MAGIC) return fail(FIVEEQ_E_INVALID, "not a live fiveeq plan");
    p->magic = 0;
    hipError_t e1 = hipGraphExecDestroy(p->exec);
    hipError_t e2 = hipGraphDestroy(p->graph);
    delete p;
    if (e1 != hipSuccess || e2 != hipSuccess) return fail(FIVEEQ_E_HIP, "graph destroy failed");
    return FIVEEQ_OK;
}

int fiveeq_hfc_conc_f64(int64_t n_members, int64_t ld, int32_t n_time, const double* e0, const double* time,
                        double* out, void* stream) {
    if (n_members < 1) return fail(FIVEEQ_E_INVALID, "n_members=%lld must be >= 1", (long long)n_members);
    if (ld < n_members) return fail(FIVEEQ_E_INVALID, "ld < n_members");
    if (n_time < 0) return fail(FIVEEQ_E_INVALID, "n_time=%d must be >= 0", n_time);
    if (n_time == 0) return FIVEEQ_OK;
    if (!e0 || !time || !out) return fail(FIVEEQ_E_INVALID, "NULL device pointer");
    const int64_t blocks = (n_members + FIVEEQ_BLOCK - 1) / FIVEEQ_BLOCK;
    if (blocks > 0x7fffffffLL) return fail(FIVEEQ_E_INVALID, "n_members too large");
    hipLaunchKernelGGL(fiveeq::hfc_conc_kernel, dim3((unsigned)blocks), dim3(FIVEEQ_BLOCK), 0, (hipStream_t)stream,
                       n_members, ld, n_time, e0, time, out);
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}

int fiveeq_stream_copy_wide_f64(int64_t n, const double* src, double* dst, void* stream) {
    if (n < 2 || (n & 1)) return fail(FIVEEQ_E_INVALID, "n=%lld must be even and >= 2", (long long)n);
    if (!src || !dst) return fail(FIVEEQ_E_INVALID, "NULL device pointer");
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return fail(FIVEEQ_E_INVALID, "pointers must be 16-byte aligned");
    const int64_t n2 = n / 2;
    const int64_t tiles = (n2 + 4 * FIVEEQ_BLOCK - 1) / (4 * FIVEEQ_BLOCK);
    const int64_t blocks = tiles < 16384 ? tiles : 16384;
    hipLaunchKernelGGL(fiveeq::stream_copy_wide_kernel, dim3((unsigned)blocks), dim3(FIVEEQ_BLOCK), 0, (hipStream_t)stream,
                       n2, reinterpret_cast<const double2*>(src), reinterpret_cast<double2*>(dst));
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}

int fiveeq_busy(int64_t iterations, double* out, void* stream) {
    if (iterations < 0 || iterations > 100000000LL) return fail(FIVEEQ_E_INVALID, "iterations=%lld outside 0..1e8", (long long)iterations);
    if (!out) return fail(FIVEEQ_E_INVALID, "NULL device pointer");
    hipLaunchKernelGGL(fiveeq::busy_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, iterations, out);
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}

int fiveeq_stream_copy_nt_f64(int64_t n, const double* src, double* dst, void* stream) {
    if (n < 1 || n % (4 * FIVEEQ_BLOCK)) return fail(FIVEEQ_E_INVALID, "n=%lld must be a positive multiple of %d", (long long)n, 4 * FIVEEQ_BLOCK);
    if (!src || !dst) return fail(FIVEEQ_E_INVALID, "NULL device pointer");
    const int64_t blocks = n / (4 * FIVEEQ_BLOCK);
    if (blocks > 0x7fffffffLL) return fail(FIVEEQ_E_INVALID, "n too large for one launch");
    hipLaunchKernelGGL(fiveeq::stream_copy_nt_kernel, dim3((unsigned)blocks), dim3(FIVEEQ_BLOCK), 0, (hipStream_t)stream, n, src, dst);
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}

int fiveeq_stream_copy_f64(int64_t n, const double* src, double* dst, void* stream) {
    if (n < 1) return fail(FIVEEQ_E_INVALID, "n=%lld must be >= 1", (long long)n);
    if (!src || !dst) return fail(FIVEEQ_E_INVALID, "NULL device pointer");
    hipLaunchKernelGGL(fiveeq::stream_copy_kernel, dim3((unsigned)(member_blocks(n) < 8192 ? member_blocks(n) : 8192)), dim3(FIVEEQ_BLOCK), 0,
                       (hipStream_t)stream, n, src, dst);
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}

}  // extern "C"

namespace {
// members per workgroup of the histogram pass: as coarse as still gives ~2048 workgroups over all rows
// (fewer, fuller flushes of the LDS histogram), a multiple of the block size
int64_t hist_chunk(int32_t n_rows, int64_t n) {
    int64_t chunk = (n * (n_rows > 0 ? n_rows : 1) + 2047) / 2048;
    if (chunk < fiveeq::HIST_CHUNK_MIN) chunk = fiveeq::HIST_CHUNK_MIN;
    return (chunk + FIVEEQ_BLOCK - 1) / FIVEEQ_BLOCK * FIVEEQ_BLOCK;
}

template <typename T>
int hist_rows(int32_t n_rows, int64_t n, int64_t ld, const T* rows, double lo, double hi, int32_t n_bins,
              uint64_t* hist, void* stream, const double* ranges = nullptr) {
    if (n_rows < 0) return fail(FIVEEQ_E_INVALID, "n_rows=%d must be >= 0", n_rows);
    if (n < 1 || ld < n) return fail(FIVEEQ_E_INVALID, "n_members=%lld, ld=%lld invalid", (long long)n, (long long)ld);
    if (n_bins < 1 || n_bins > fiveeq::HIST_MAX_BINS)
        return fail(FIVEEQ_E_INVALID, "n_bins=%d outside 1..%d", n_bins, fiveeq::HIST_MAX_BINS);
    if (!ranges && (!(hi > lo) || !std::isfinite(lo) || !std::isfinite(hi))) return fail(FIVEEQ_E_INVALID, "need finite lo < hi");
    if (n_rows == 0) return FIVEEQ_OK;
    if (!rows || !hist) return fail(FIVEEQ_E_INVALID, "NULL device pointer");
    if (n_rows > 65535) return fail(FIVEEQ_E_INVALID, "n_rows=%d exceeds the 65535 rows of one launch", n_rows);
    int64_t chunk = hist_chunk(n_rows, n);
    if (ranges) {
        // the summary's histogram (a few rows): every workgroup zeroes and flushes all n_bins counters, which at 2048 workgroups
        // of 18k members each is a third of the pass (68 us for 3 x 12.5M fp32 values at 4096 bins, 40 at 1024 bins) — at least
        // 8 members per bin and workgroup: 47 us.  (The ring passes count 64+ rows per launch and are far beyond that already.)
        const int64_t floor_ = (8LL * n_bins + FIVEEQ_BLOCK - 1) / FIVEEQ_BLOCK * FIVEEQ_BLOCK;
        if (chunk < floor_) chunk = floor_;
    }
    const int64_t chunks = (n + chunk - 1) / chunk;
    if (chunks > 0x7fffffffLL) return fail(FIVEEQ_E_INVALID, "n_members too large");
    const double inv_w = ranges ? 0.0 : (double)n_bins / (hi - lo);
    hipLaunchKernelGGL(fiveeq::hist_rows_kernel<T>, dim3((unsigned)chunks, (unsigned)n_rows), dim3(FIVEEQ_BLOCK), 0,
                       (hipStream_t)stream, n, ld, chunk, rows, lo, inv_w, n_bins, reinterpret_cast<unsigned long long*>(hist),
                       ranges);
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}

template <typename T>
int math_probe(int32_t op, int64_t n, const T* x, T* y, void* stream) {
    const bool packed_op = sizeof(T) == 4 && op >= 8 && op <= 12;          // fp32: the packed twin on element pairs
    if ((op < 0 || op > 4) && !packed_op) return fail(FIVEEQ_E_INVALID, "op=%d outside 0..4 (fp32: also 8..12)", op);
    if (n < 1) return fail(FIVEEQ_E_INVALID, "n=%lld must be >= 1", (long long)n);
    if (packed_op && (n & 1)) return fail(FIVEEQ_E_INVALID, "packed ops need an even n");
    if (!x || !y) return fail(FIVEEQ_E_INVALID, "NULL device pointer");
    const int64_t blocks = member_blocks(n);
    if (blocks > 0x7fffffffLL) return fail(FIVEEQ_E_INVALID, "n too large");
    hipLaunchKernelGGL(fiveeq::math_probe_kernel<T>, dim3((unsigned)blocks), dim3(FIVEEQ_BLOCK), 0, (hipStream_t)stream,
                       op, n, x, y);
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}
}  // namespace

extern "C" {
int fiveeq_hist_rows_f64(int32_t n_rows, int64_t n_members, int64_t ld, const double* rows, double lo, double hi,
                         int32_t n_bins, uint64_t* hist, void* stream) {
    return hist_rows<double>(n_rows, n_members, ld, rows, lo, hi, n_bins, hist, stream);
}
int fiveeq_hist_bins(int32_t n_rows, int64_t n_members, int64_t ld, const uint16_t* bins, int32_t n_bins, uint64_t* hist,
                     void* stream) {
    if (n_rows < 0) return fail(FIVEEQ_E_INVALID, "n_rows=%d must be >= 0", n_rows);
    if (n_members < 1 || ld < n_members)
        return fail(FIVEEQ_E_INVALID, "n_members=%lld, ld=%lld invalid", (long long)n_members, (long long)ld);
    if (n_bins < 1 || n_bins > fiveeq::HIST_MAX_BINS)
        return fail(FIVEEQ_E_INVALID, "n_bins=%d outside 1..%d", n_bins, fiveeq::HIST_MAX_BINS);
    if (n_rows == 0) return FIVEEQ_OK;
    if (!bins || !hist) return fail(FIVEEQ_E_INVALID, "NULL device pointer");
    if (n_rows > 65535) return fail(FIVEEQ_E_INVALID, "n_rows=%d exceeds the 65535 rows of one launch", n_rows);
    int64_t chunk = hist_chunk(n_rows, n_members);
    chunk = (chunk + 8 * FIVEEQ_BLOCK - 1) / (8 * FIVEEQ_BLOCK) * (8 * FIVEEQ_BLOCK);      // eight members per lane and load
    const int64_t chunks = (n_members + chunk - 1) / chunk;
    if (chunks > 0x7fffffffLL) return fail(FIVEEQ_E_INVALID, "n_members too large");
    hipLaunchKernelGGL(fiveeq::hist_bins_kernel, dim3((unsigned)chunks, (unsigned)n_rows), dim3(FIVEEQ_BLOCK), 0,
                       (hipStream_t)stream, n_members, ld, chunk, bins, n_bins, reinterpret_cast<unsigned long long*>(hist));
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}
int fiveeq_hist_rows_f32(int32_t n_rows, int64_t n_members, int64_t ld, const float* rows, double lo, double hi,
                         int32_t n_bins, uint64_t* hist, void* stream) {
    return hist_rows<float>(n_rows, n_members, ld, rows, lo, hi, n_bins, hist, stream);
}
// ---- end-of-run summary passes (kernels 6a-6c) ----------------------------------------------------------------
}  // extern "C"
namespace {
// members per workgroup of a summary pass: the histogram pass's chunking, rounded to whole 16-byte-per-lane loads
int64_t summary_chunk(int32_t n_rows, int64_t n) {
    const int64_t unit = 4 * FIVEEQ_BLOCK;                  // 4 floats / 2 doubles per lane: a multiple of both
    return (hist_chunk(n_rows, n) + unit - 1) / unit * unit;
}
int summary_check(int32_t n_rows, int64_t n, int64_t ld, const void* rows) {
    if (n_rows < 0) return fail(FIVEEQ_E_INVALID, "n_rows=%d must be >= 0", n_rows);
    if (n_rows > 65535) return fail(FIVEEQ_E_INVALID, "n_rows=%d exceeds the 65535 rows of one launch", n_rows);
    if (n < 1 || ld < n) return fail(FIVEEQ_E_INVALID, "n_members=%lld, ld=%lld invalid", (long long)n, (long long)ld);
    if (n_rows > 0 && !rows) return fail(FIVEEQ_E_INVALID, "NULL device pointer");
    return FIVEEQ_OK;
}
template <typename T>
int row_moments(int32_t n_rows, int64_t n, int64_t ld, const T* rows, double* partial, double* moments, void* stream) {
    if (int rc = summary_check(n_rows, n, ld, rows)) return rc;
    if (n_rows == 0) return FIVEEQ_OK;
    if (!partial || !moments) return fail(FIVEEQ_E_INVALID, "NULL device pointer");
    const int64_t chunk = summary_chunk(n_rows, n);
    const int64_t chunks = (n + chunk - 1) / chunk;
    if (chunks > 0x7fffffffLL) return fail(FIVEEQ_E_INVALID, "n_members too large");
    hipLaunchKernelGGL(fiveeq::row_moments_kernel<T>, dim3((unsigned)chunks, (unsigned)n_rows), dim3(FIVEEQ_BLOCK), 0,
                       (hipStream_t)stream, n, ld, chunk, rows, partial);
    hipLaunchKernelGGL(fiveeq::row_moments_fold_kernel, dim3((unsigned)n_rows), dim3(64), 0, (hipStream_t)stream, chunks,
                       partial, moments);
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}
template <typename T>
int select_bins(int32_t n_rows, int64_t n, int64_t ld, const T* rows, const double* ranges, int32_t n_bins,
                const uint32_t* binmask, T* cand, int64_t cap, uint64_t* cand_n, void* stream) {
    if (int rc = summary_check(n_rows, n, ld, rows)) return rc;
    if (n_bins < 1 || n_bins > fiveeq::HIST_MAX_BINS)
        return fail(FIVEEQ_E_INVALID, "n_bins=%d outside 1..%d", n_bins, fiveeq::HIST_MAX_BINS);
    if (cap < 0) return fail(FIVEEQ_E_INVALID, "cap=%lld must be >= 0", (long long)cap);
    if (n_rows == 0) return FIVEEQ_OK;
    if (!ranges || !binmask || !cand_n || (cap > 0 && !cand)) return fail(FIVEEQ_E_INVALID, "NULL device pointer");
    const int64_t chunk = summary_chunk(n_rows, n);
    const int64_t chunks = (n + chunk - 1) / chunk;
    if (chunks > 0x7fffffffLL) return fail(FIVEEQ_E_INVALID, "n_members too large");
    hipLaunchKernelGGL(fiveeq::select_bins_kernel<T>, dim3((unsigned)chunks, (unsigned)n_rows), dim3(FIVEEQ_BLOCK), 0,
                       (hipStream_t)stream, n, ld, chunk, rows, ranges, n_bins, binmask, cand, cap,
                       reinterpret_cast<unsigned long long*>(cand_n));
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}
template <typename T>
int select_pick(int32_t n_rows, int32_t n_seg, int64_t width, const T* pool, const uint64_t* seg_n, int32_t n_targets,
                const int64_t* ranks, double* picked, void* stream) {
    if (n_rows < 0) return fail(FIVEEQ_E_INVALID, "n_rows=%d invalid", n_rows);
    if (n_seg < 1) return fail(FIVEEQ_E_INVALID, "n_seg=%d must be >= 1", n_seg);
    if (width < 0) return fail(FIVEEQ_E_INVALID, "width=%lld must be >= 0", (long long)width);
    if (n_targets < 1 || n_targets > 65535) return fail(FIVEEQ_E_INVALID, "n_targets=%d outside 1..65535", n_targets);
    if (n_rows == 0) return FIVEEQ_OK;
    if ((width > 0 && !pool) || !seg_n || !ranks || !picked) return fail(FIVEEQ_E_INVALID, "NULL device pointer");
    hipLaunchKernelGGL(fiveeq::select_pick_kernel<T>, dim3((unsigned)n_rows, (unsigned)n_targets), dim3(fiveeq::PICK_BLOCK), 0,
                       (hipStream_t)stream, n_seg, width, pool, reinterpret_cast<const unsigned long long*>(seg_n), n_targets,
                       reinterpret_cast<const long long*>(ranks), picked);
    HIP_TRY(hipGetLastError());
    return FIVEEQ_OK;
}
}  // namespace
extern "C" {
int fiveeq_select_pick_f64(int32_t n_rows, int32_t n_seg, int64_t width, const double* pool, const uint64_t* seg_n,
                           int32_t n_targets, const int64_t* ranks, double* picked, void* stream) {
    return select_pick<double>(n_rows, n_seg, width, pool, seg_n, n_targets, ranks, picked, stream);
}
int fiveeq_select_pick_f32(int32_t n_rows, int32_t n_seg, int64_t width, const float* pool, const uint64_t* seg_n,
                           int32_t n_targets, const int64_t* ranks, double* picked, void* stream) {
    return select_pick<float>(n_rows, n_seg, width, pool, seg_n, n_targets, ranks, picked, stream);
}
int64_t fiveeq_row_moments_chunks(int32_t n_rows, int64_t n_members) {
    if (n_rows < 1 || n_members < 1) return 0;
    const int64_t chunk = summary_chunk(n_rows, n_members);
    return (n_members + chunk - 1) / chunk;
}
int fiveeq_row_moments_f64(int32_t n_rows, int64_t n_members, int64_t ld, const double* rows, double* partial,
                           double* moments, void* stream) {
    return row_moments<double>(n_rows, n_members, ld, rows, partial, moments, stream);
}
int fiveeq_row_moments_f32(int32_t n_rows, int64_t n_members, int64_t ld, const float* rows, double* partial,
                           double* moments, void* stream) {
    return row_moments<float>(n_rows, n_members, ld, rows, partial, moments, stream);
}
int fiveeq_hist_rows_ranged_f64(int32_t n_rows, int64_t n_members, int64_t ld, const double* rows, const double* ranges,
                                int32_t n_bins, uint64_t* hist, void* stream) {
    if (!ranges) return fail(FIVEEQ_E_INVALID, "ranges is NULL");
    return hist_rows<double>(n_rows, n_members, ld, rows, 0.0, 0.0, n_bins, hist, stream, ranges);
}
int fiveeq_hist_rows_ranged_f32(int32_t n_rows, int64_t n_members, int64_t ld, const float* rows, const double* ranges,
                                int32_t n_bins, uint64_t* hist, void* stream) {
    if (!ranges) return fail(FIVEEQ_E_INVALID, "ranges is NULL");
    return hist_rows<float>(n_rows, n_members, ld, rows, 0.0, 0.0, n_bins, hist, stream, ranges);
}
int fiveeq_select_bins_f64(int32_t n_rows, int64_t n_members, int64_t ld, const double* rows, const double* ranges,
                           int32_t n_bins, const uint32_t* binmask, double* cand, int64_t cap, uint64_t* cand_n, void* stream) {
    return select_bins<double>(n_rows, n_members, ld, rows, ranges, n_bins, binmask, cand, cap, cand_n, stream);
}
int fiveeq_select_bins_f32(int32_t n_rows, int64_t n_members, int64_t ld, const float* rows, const double* ranges,
                           int32_t n_bins, const uint32_t* binmask, float* cand, int64_t cap, uint64_t* cand_n, void* stream) {
    return select_bins<float>(n_rows, n_members, ld, rows, ranges, n_bins, binmask, cand, cap, cand_n, stream);
}
int fiveeq_math_probe_f64(int32_t op, int64_t n, const double* x, double* y, void* stream) {
    return math_probe<double>(op, n, x, y, stream);
}
int fiveeq_math_probe_f32(int32_t op, int64_t n, const float* x, float* y, void* stream) {
    return math_probe<float>(op, n, x, y, stream);
}

}  // extern "C"
