// fiveeq_device.hpp — gfx950 device code for the five-equation FaIR ensemble step.
//
// Written for MI355X (CDNA4, wave64) only.  The path is element-wise over ensemble
// members: one member per lane, struct-of-arrays rows so that a wave's 64 lanes read
// 64 consecutive elements of each row (512 B per wave-instruction at fp64), no MFMA.
//
// The reference (stujen/fiveEqSCM @ v0) has no implementation of these equations
// (only `emissions[0]*exp(-time)`, U_FaIR/concentrations.py:4-5); the function
// split below follows the names it reserves at .coveragerc:12-19
// (alpha_val, step_conc, step_forc, step_temp).  See include/fiveeq.h for the model.
//
// Kernels in this file:
//   1   step_kernel        one timestep per launch — the north-star form; HBM-bound, A = w(2SP + 4G + 7) per member-step
//   2   fused_kernel       time-fused (and, INV = true, concentration-driven); state in registers; VALU-bound
//   2c  small_kernel       small ensembles: one member per quad of lanes (pool per lane), the model in registers
//   3   hfc_conc_kernel    the reference's one function over an ensemble
//   4   hist_rows_kernel   fixed-bin histograms (+ moments) of rows: the pass of the streamed histogram pipelines
//   5   lhs_kernel         shard-computable Latin hypercube (keyed Feistel bijection)
//   diagnostics: stream_copy_kernel, stream_copy_wide_kernel, stream_copy_nt_kernel, math_probe_kernel, busy_kernel
// All model arithmetic is member_step(): every kernel that steps the model gives the same bits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Floating-point contraction is OFF for this translation unit (here and in csrc/Makefile): the compiler never
// decides which a*b+c fuse.  Every fused multiply-add of the model step is written as fe_fma() below, so
// "per-step == fused == K-step == small == graph bit for bit" and the distance to the CPU oracle are properties of this
// source, not of a hipcc release.
#pragma clang fp contract(off)

#ifndef FIVEEQ_BLOCK
#define FIVEEQ_BLOCK 256          // threads per workgroup of the fused / inverse / utility kernels (4 waves)
#endif
#ifndef FIVEEQ_STEP_BLOCK
#define FIVEEQ_STEP_BLOCK 64      // threads per workgroup of the per-step kernel: ONE wave (measured best, below)
#endif
// (Experiments that lost — device-library math, non-temporal trajectory stores, model constants in SGPRs —
// are recorded in profiles/r01/ab_variants.txt; their code paths are gone.)
#ifndef FIVEEQ_SMALL_BLOCK
#define FIVEEQ_SMALL_BLOCK 256    // threads per workgroup of the small-ensemble kernel: four waves, one per SIMD of a CU (measured, below)
#endif
#ifndef FIVEEQ_FUSED_CHUNK
#define FIVEEQ_FUSED_CHUNK 125    // drive-table steps staged into LDS per refill (fused kernel)
#endif

namespace fiveeq {

constexpr int MAX_GAS = 3;
constexpr int MAX_POOLS = 4;
constexpr int DRIVE_STRIDE = 8;

// ---------------------------------------------------------------------------------
// Shared model in kernel precision, passed BY VALUE as the FIRST kernel argument (496 B): it
// lands at offset 0 of the kernarg segment, from where each workgroup stages it into LDS once
// (stage_model below); lanes then read it with broadcast ds_reads.  No HBM traffic per member.
// ---------------------------------------------------------------------------------
template <typename T>
struct KGas {
    T ndt_over_tau[MAX_POOLS];  // -dt / tau_i
    T atc[MAX_POOLS];           // a_i * tau_i * c      (so x_eq_i = atc_i * E * alpha)
    T g0, inv_g1, ra, inv_c, C0, inv_C0, sqrtC0, f1, f2, f3;
};
template <typename T>
struct KModel {
    KGas<T> gas[MAX_GAS];
    T em1_d[2];                 // expm1(-dt/d_j), computed on the host in fp64
    T iirf_max;
    T dt;
};

__device__ __forceinline__ double fe_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fe_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---------------------------------------------------------------------------------
// Lane value types.  A lane carries ONE member (V = double or float) or, in the packed fp32 kernels, TWO CONSECUTIVE
// members (V = float2v: lane l of a wave owns members 2l and 2l + 1 of the wave's 128).  Packed lanes load and store
// 8 bytes per row (512 B per wave-instruction, the fp64 kernels' access shape) and their multiplies, adds and FMAs issue
// as v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 — one instruction for both members (5.0-5.2 cycles per wave-instruction
// against 2 x 2.9-3.6 for the scalar forms, profiles/r03/valu_rates_microbench.txt); what the ISA has no packed form
// for (v_rcp_f32, v_sqrt_f32, v_rndne_f32, v_ldexp_f32, v_frexp_*, min/max, compares and selects) runs per component.
// Packed arithmetic is IEEE per component and every routine below mirrors its scalar twin operation by operation, so a
// member's result does not depend on which kernel shape computed it (tested bit for bit).
// ---------------------------------------------------------------------------------
typedef float float2v __attribute__((ext_vector_type(2)));

template <typename V> struct Lane { using S = V; static constexpr int W = 1; };
template <> struct Lane<float2v> { using S = float; static constexpr int W = 2; };

__device__ __forceinline__ float2v fe_fma(float2v a, float2v b, float2v c) { return __builtin_elementwise_fma(a, b, c); }
// fma with any mix of lane values and shared scalars (constants are splat into both components)
template <typename V, typename A, typename B, typename C>
__device__ __forceinline__ V fma3(A a, B b, C c) { return fe_fma((V)a, (V)b, (V)c); }

// per-lane predicates and selects
struct Mask2 { bool x, y; };
template <typename V> struct MaskOf { using type = bool; };
template <> struct MaskOf<float2v> { using type = Mask2; };
__device__ __forceinline__ bool fe_gt0(double v) { return v > 0.0; }
__device__ __forceinline__ bool fe_gt0(float v) { return v > 0.0f; }
__device__ __forceinline__ Mask2 fe_gt0(float2v v) { return Mask2{v.x > 0.0f, v.y > 0.0f}; }
__device__ __forceinline__ double fe_sel(bool m, double a, double b) { return m ? a : b; }
__device__ __forceinline__ float fe_sel(bool m, float a, float b) { return m ? a : b; }
__device__ __forceinline__ float2v fe_sel(Mask2 m, float2v a, float2v b) { return float2v{m.x ? a.x : b.x, m.y ? a.y : b.y}; }

template <int P0, int P1, int P2>
struct Layout {
    static constexpr int G = (P0 > 0) + (P1 > 0) + (P2 > 0);
    static constexpr int SP = P0 + P1 + P2;
    __host__ __device__ static constexpr int pools(int g) { return g == 0 ? P0 : (g == 1 ? P1 : P2); }
    __host__ __device__ static constexpr int off(int g) { return g == 0 ? 0 : (g == 1 ? P0 : P0 + P1); }
};

// ---------------------------------------------------------------------------------
// Math.  Every transcendental of the step is written for the argument range the model can
// produce (each pinned to <= 2 ulp against a CPU libm through fiveeq_math_probe_*).  expm1 is the
// hot one (one per pool per member-step), always with an argument <= 0:
//   x = k ln2 + r, |r| <= ln2/2 ;  expm1(x) = 2^k (expm1 r) + (2^k - 1)
// with expm1(r) = r + r^2 Q(r), Q a degree-10 near-minimax polynomial (8.5e-19 relative).  For
// k = 0 the result is expm1(r) itself, so small arguments (the tau ~ 1e6 yr pool: x ~ -1e-6)
// keep full RELATIVE accuracy.
// ---------------------------------------------------------------------------------
// expm1(r) on |r| <= ln2/2 as r + r^2 Q(r): shared by fe_expm1_neg and fe_exp.  Q is the degree-10
// interpolant of (expm1(r) - r)/r^2 at the Chebyshev nodes of the interval (computed in 60-digit
// decimal arithmetic, coefficients rounded to double): approximation error 8.5e-19 relative, where
// the Taylor polynomial needs degree 12 for 1.2e-17.  Two fewer FMAs on each of the nine exp-type
// calls of a three-gas member-step.
__device__ __forceinline__ double fe_expm1_reduced(double r) {
    double q = 0x1.1f72fc730b4ffp-29;
    q = __builtin_fma(q, r, 0x1.af4ddd84882fep-26);
    q = __builtin_fma(q, r, 0x1.27e4db67b4303p-22);
    q = __builtin_fma(q, r, 0x1.71de02375656cp-19);
    q = __builtin_fma(q, r, 0x1.a01a01a6d7808p-16);
    q = __builtin_fma(q, r, 0x1.a01a01abe62ddp-13);
    q = __builtin_fma(q, r, 0x1.6c16c16c162d6p-10);
    q = __builtin_fma(q, r, 0x1.11111111100dfp-7);
    q = __builtin_fma(q, r, 0x1.5555555555556p-5);
    q = __builtin_fma(q, r, 0x1.5555555555557p-3);
    q = __builtin_fma(q, r, 0x1.0000000000000p-1);
    return __builtin_fma(r * r, q, r);
}
// x = k ln2 + r with |r| <= ln2/2 (two-step Cody-Waite; ln2 hi has 32 zero low bits)
__device__ __forceinline__ double fe_reduce_ln2(double x, double& k) {
    k = __builtin_rint(x * 1.4426950408889634);              // v_rndne_f64
    const double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);
    return __builtin_fma(-k, 1.90821492927058770002e-10, r);
}

__device__ __forceinline__ double fe_expm1_neg(double x) {
    x = fmax(x, -800.0);                                     // exp(-800) == 0: result -1
    double k;
    const double p = fe_expm1_reduced(fe_reduce_ln2(x, k));
    const double s = __builtin_ldexp(1.0, (int)k);           // 2^k, k <= 0
    return __builtin_fma(s, p, s - 1.0);                     // k = 0: exactly p
}
// fp32 routines.  Same scheme, re-cut for what the fp32 VALU is good at (round 3; each step measured on the fused
// config-5 shard, profiles/r03/ab_variants.txt):
//   * expm1(r) = r + r^2 Q(r) with Q the DEGREE-4 interpolant of (expm1(r) - r)/r^2 at the Chebyshev nodes of
//     |r| <= ln2/2 (2.3e-8 relative — the degree-5 Taylor polynomial it replaces had 1.8e-8 — one FMA fewer per call);
//   * the reduction x = k ln2 + r takes k from the magic-number add u = fma(x, log2 e, 1.5 * 2^23) (round to nearest even
//     in the add itself), k = u - magic, and builds 2^k from u's low mantissa bits with one integer shift-add — no
//     v_rndne / v_cvt / v_ldexp; the argument is clamped at -87 so that 2^k stays a normal float (expm1 is -1 below -17);
//   * exp (the alpha closure) uses the hardware 2^t (v_exp_f32, 1 ulp) on t = x log2(e) with the product's rounding
//     error and the low part of log2(e) folded back in: exp(x) = 2^t (1 + lo ln2), six instructions instead of fourteen.
// All within 2 ulp(float) of libm over the model's ranges (tests/test_engine_gpu.py, through fiveeq_math_probe_f32).
__device__ __forceinline__ float fe_expm1_reduced(float r) {
    float q = 0x1.6d10fcp-10f;
    q = __builtin_fmaf(q, r, 0x1.120b62p-7f);
    q = __builtin_fmaf(q, r, 0x1.55551ap-5f);
    q = __builtin_fmaf(q, r, 0x1.5554dep-3f);
    q = __builtin_fmaf(q, r, 0.5f);
    return __builtin_fmaf(r * r, q, r);
}
constexpr float F32_LOG2E = 1.44269504088896341f;
constexpr float F32_LN2 = 0.693147182f;
constexpr float F32_RINT_MAGIC = 12582912.0f;                // 1.5 * 2^23
// 2^k for the integer k held in the low mantissa bits of u = 1.5 * 2^23 + k, -126 <= k <= 0: (bits(u) << 23) + bits(1.0f),
// one v_lshl_add_u32.  Written as inline asm: as plain C++ the packed form below was MISCOMPILED by hipcc 7.2 (the shift-add
// of the second component was dropped and the first component's 2^k used for both members; found by the packed-vs-scalar
// probe test).  Not volatile: the scheduler may still move it.
__device__ __forceinline__ float fe_exp2_from_magic(float u) {
    float s;
    asm("v_lshl_add_u32 %0, %1, 23, 1.0" : "=v"(s) : "v"(u));
    return s;
}
__device__ __forceinline__ float fe_expm1_neg(float x) {
    x = fmaxf(x, -87.0f);                                    // k >= -126
    const float u = __builtin_fmaf(x, F32_LOG2E, F32_RINT_MAGIC);      // magic + rint(x log2 e)
    const float k = u - F32_RINT_MAGIC;
    // ONE fma for the reduction: ln2's own rounding error enters the result as 2^k |k| 2^-26 <= 1e-8 absolute on a result of
    // magnitude >= 0.29 whenever k != 0 (and not at all for k = 0): 0.3 ulp at worst, where exp() proper would need the
    // two-step Cody-Waite form.  Same 1.01 ulp maximum over the probe ranges; one instruction fewer on each of six calls.
    const float r = __builtin_fmaf(-k, F32_LN2, x);
    const float p = fe_expm1_reduced(r);
    const float s = fe_exp2_from_magic(u);
    return __builtin_fmaf(s, p, s - 1.0f);                   // k = 0: exactly p
}

// exp(x) for the alpha closure.  The argument is clamped to +-700 so that alpha is always a
// finite normal number (e^+-700 ~ 1e+-304) and the Newton reciprocal below is always valid.
__device__ __forceinline__ double fe_exp(double x) {
    x = fmin(fmax(x, -700.0), 700.0);
    double k;
    const double p = fe_expm1_reduced(fe_reduce_ln2(x, k));
    return __builtin_ldexp(1.0 + p, (int)k);
}
constexpr float F32_LOG2E_HI = 0x1.715476p+0f;               // log2(e) rounded to float, and what it leaves
constexpr float F32_LOG2E_LO = 0x1.4ae0cp-26f;
__device__ __forceinline__ float fe_exp(float x) {
    x = fminf(fmaxf(x, -80.0f), 80.0f);                      // alpha stays a finite normal float
    const float t = x * F32_LOG2E_HI;
    float lo = __builtin_fmaf(x, F32_LOG2E_HI, -t);          // the product's rounding error, exactly
    lo = __builtin_fmaf(x, F32_LOG2E_LO, lo);
    const float e = __builtin_amdgcn_exp2f(t);               // v_exp_f32
    return __builtin_fmaf(e, lo * F32_LN2, e);
}

// 1/a for finite normal a > 0 (alpha): v_rcp_f64 seed + two Newton steps (<= 1 ulp), without the
// scale / fixup sequence a full IEEE division needs for subnormal and infinite operands.
__device__ __forceinline__ double fe_rcp(double a) {
    double y = __builtin_amdgcn_rcp(a);
    double e = __builtin_fma(-a, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-a, y, 1.0);
    return __builtin_fma(y, e, y);
}
__device__ __forceinline__ float fe_rcp(float a) {
    return __builtin_amdgcn_rcpf(a);                         // v_rcp_f32: 1 ulp.  (A Newton step on top, <= 0.5 ulp, was 2 % of the
}                                                            // fused fp32 kernel and moved no fp32-vs-fp64 figure: r03/ab_variants.txt)

// ln(x) for finite normal x > 0 (a concentration ratio).  The classic fdlibm scheme:
// x = 2^k (1+f) with sqrt(1/2) <= 1+f < sqrt(2);  s = f/(2+f);  ln(1+f) = f - f^2/2 + s (f^2/2 + R(s^2))
// with R the degree-7 minimax polynomial in s^2 (Lg1..Lg7, |error| < 2^-58.45), and k ln2 added in
// hi/lo parts.  The quotient uses the Newton reciprocal (2+f is in [1.7, 2.42]).  ~35 VALU ops against
// ~95 for the general device-library routine, which carries double-double arithmetic and
// special-case selects this argument range never needs.
__device__ __forceinline__ double fe_log(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);                   // [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;                                         // [sqrt(1/2), sqrt(2))
    k = low ? k - 1 : k;
    const double dk = (double)k;
    const double f = m - 1.0;
    const double s = f * fe_rcp(2.0 + f);
    const double z = s * s;
    const double w = z * z;
    double t1 = __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01);   // Lg6, Lg4
    t1 = __builtin_fma(w, t1, 3.999999999940941908e-01);                                 // Lg2
    t1 = w * t1;
    double t2 = __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01);   // Lg7, Lg5
    t2 = __builtin_fma(w, t2, 2.857142874366239149e-01);                                 // Lg3
    t2 = __builtin_fma(w, t2, 6.666666666666735130e-01);                                 // Lg1
    const double R = __builtin_fma(z, t2, t1);
    const double hfsq = 0.5 * f * f;
    const double tail = __builtin_fma(dk, 1.90821492927058770002e-10, s * (hfsq + R));  // + k ln2_lo
    return __builtin_fma(dk, 6.93147180369123816490e-01, -((hfsq - tail) - f));          // k ln2_hi - ...
}
// fp32 log: ln(x) = ln2 * log2(x) with the hardware log2 (v_log_f32).  Measured on gfx950 (tools/microbench/hw_log_accuracy.hip,
// profiles/r03/hw_log_accuracy.txt): v_log_f32 is within 1 ulp of log2(x) over [0.5, 16] AND right next to 1 (x in
// [1, 1 + 1e-5]: 0.92 ulp of a result of ~1e-6 — no loss of relative accuracy where ln x -> 0, which is where the CO2 forcing
// starts), and it returns exactly 0 at x = 1.  The product with ln2 = hi + lo carries the rounding error of t * hi along:
// <= 2.1 ulp of ln(x), mean 0.55.  Five instructions per member where the frexp + division + polynomial form above took
// twenty (fdlibm's, < 1 ulp): the forcing's log was 9 % of the fused fp32 kernel (r03/ab_variants.txt section 16).
constexpr float F32_LN2_H = 0x1.62e430p-1f;                  // ln2 rounded to float, and what it leaves
constexpr float F32_LN2_L = -0x1.05c610p-29f;
__device__ __forceinline__ float fe_log(float x) {
    const float t = __builtin_amdgcn_logf(x);
    const float p = t * F32_LN2_H;
    const float e = __builtin_fmaf(t, F32_LN2_H, -p);
    return p + __builtin_fmaf(t, F32_LN2_L, e);
}

// sqrt(x) for finite normal x > 0 (a concentration): v_rsq_f64 seed, one Goldschmidt step and a
// final residual correction (<= 1 ulp; a second Goldschmidt step was redundant), without the rescaling a
// full sqrt needs near the ends of the exponent range.  (Also tried in round 2 and not kept: magic-number
// rint + integer-built 2^k in the exp core — 6 fewer VALU per step but +4 VGPRs: 7 -> 6 waves/SIMD in the
// per-step kernel.)
__device__ __forceinline__ double fe_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);                // >= 23 good bits
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);              // one Goldschmidt step: ~2^-45
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);                // residual (Newton) correction: quadratic again
    return __builtin_fma(d, h, g);
}
__device__ __forceinline__ float fe_sqrt(float x) {
    return __builtin_amdgcn_sqrtf(x);                        // v_sqrt_f32: 1 ulp for normal x > 0
}
__device__ __forceinline__ double fe_min(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float fe_min(float a, float b) { return fminf(a, b); }

// ---- packed fp32 twins (two members per lane): the same operations in the same order as the float routines above ----
__device__ __forceinline__ float2v fe_min(float2v a, float b) { return float2v{fminf(a.x, b), fminf(a.y, b)}; }
__device__ __forceinline__ float2v fe_expm1_reduced(float2v r) {
    float2v q = (float2v)0x1.6d10fcp-10f;
    q = fe_fma(q, r, (float2v)0x1.120b62p-7f);
    q = fe_fma(q, r, (float2v)0x1.55551ap-5f);
    q = fe_fma(q, r, (float2v)0x1.5554dep-3f);
    q = fe_fma(q, r, (float2v)0.5f);
    return fe_fma(r * r, q, r);
}
__device__ __forceinline__ float2v fe_expm1_neg(float2v x) {
    x = float2v{fmaxf(x.x, -87.0f), fmaxf(x.y, -87.0f)};
    const float2v u = fe_fma(x, (float2v)F32_LOG2E, (float2v)F32_RINT_MAGIC);
    const float2v k = u - F32_RINT_MAGIC;
    const float2v r = fe_fma(-k, (float2v)F32_LN2, x);
    const float2v p = fe_expm1_reduced(r);
    const float2v s = float2v{fe_exp2_from_magic(u.x), fe_exp2_from_magic(u.y)};
    return fe_fma(s, p, s - 1.0f);
}
__device__ __forceinline__ float2v fe_exp(float2v x) {
    x = float2v{fminf(fmaxf(x.x, -80.0f), 80.0f), fminf(fmaxf(x.y, -80.0f), 80.0f)};
    const float2v t = x * F32_LOG2E_HI;
    float2v lo = fe_fma(x, (float2v)F32_LOG2E_HI, -t);
    lo = fe_fma(x, (float2v)F32_LOG2E_LO, lo);
    const float2v e = float2v{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    return fe_fma(e, lo * F32_LN2, e);
}
__device__ __forceinline__ float2v fe_rcp(float2v a) {
    return float2v{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)};
}
__device__ __forceinline__ float2v fe_log(float2v x) {
    const float2v t = float2v{__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)};
    const float2v p = t * F32_LN2_H;
    const float2v e = fe_fma(t, (float2v)F32_LN2_H, -p);
    return p + fe_fma(t, (float2v)F32_LN2_L, e);
}
__device__ __forceinline__ float2v fe_sqrt(float2v x) {
    return float2v{__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)};
}

// ---------------------------------------------------------------------------------
// One member, one step.  All state lives in registers; the caller moves it.
//   drv : this step's drive record (LDS): [0..2] E_g, [3..5] cumE_g, [6] F_ext
//   rr  : per-member r0,rC,rT per gas ;  qq: per-member q_1,q_2
//   R,S : in/out ;  C[g], Tnew: outputs
// Every loop has compile-time bounds and is fully unrolled: arrays stay in VGPRs.
//
// INV = true is the concentration-driven (inverse) form: drv[g] holds the TARGET concentration
// at the end of the step, the member's cumulative emissions cum[g] are per-member state, and
// the emission rate that reaches the target is diagnosed from the same pool equations
//     C* - C0 = sum_i R_i (1 + em1_i) - E alpha sum_i (a_i tau_i c) em1_i
// and returned in out[g]; the pools are then advanced with that E.
// ---------------------------------------------------------------------------------
// V is the lane value type (double, float, or float2v = two members per lane); S its scalar type: the shared model and
// the drive record are S, everything per member is V.
//
// COMP = true is the COMPENSATED fp32 form (round 6; opt-in, register-resident kernels only — fiveeq_run_fused_comp_f32):
//   * every POOL carries a second word (Rlo): the rounding error of its own update, fed back into the next one —
//     y = fma(em1, x, lo); t = R + y; lo = y - (t - R); R = t (Kahan's summation with the product fused into the first add);
//     three more instructions per pool and step, no HBM bytes (the words live and die in registers).  The thermal boxes are NOT
//     compensated: with the forcing below their rounding is 4e-7 of T, and their six instructions were 2.5 % of the kernel;
//   * the forcing is computed from the EXCESS sumN = C - C0 instead of from the rounded C: ln(C/C0) = log1p(x), x = sumN / C0, as
//     ln(u) + (x - (u - 1)) with u = fl(1 + x) (the correction's own 1/u is dropped: it matters only where u ~ 1, where it is 1),
//     and sqrt C - sqrt C0 = sumN / (sqrt C + sqrt C0).  In fp32 the default form loses the small excess of the first decades to
//     the rounding of C itself (ulp(278 ppm) = 3e-5 ppm against an excess of 1e-2 ppm: a forcing good to 1e-3 relative) — that,
//     not the state, is what bounds T in fp32.
// Against 50-digit arithmetic over the 24 golden members: C 2.9e-6 -> 1.4e-7, T 1.7e-5 -> 7e-7 (profiles/r06/fp32_compensated.txt).
// It is its own arithmetic: NOT bit-identical to the default forms, and the per-step kernels (state in HBM) do not have it.
template <typename V, typename L, int g, bool INV, bool COMP = false>
__device__ __forceinline__ V gas_step(const KModel<typename Lane<V>::S>& km, const KGas<typename Lane<V>::S>& kg,
                                      const typename Lane<V>::S* __restrict__ drv, const V (&rr)[3 * L::G], const V T_old,
                                      V (&R)[L::SP], V (&out)[L::G], V (&cum)[L::G], V (&Rlo)[L::SP]) {
    using S = typename Lane<V>::S;
    static_assert(!INV || Lane<V>::W == 1, "the concentration-driven form has no packed instantiation");
    static_assert(!COMP || (!INV && sizeof(S) == 4), "the compensated form is an fp32 form of the emission-driven step");
    constexpr int P = L::pools(g);
    constexpr int o = L::off(g);
    // --- alpha_val -----------------------------------------------------------------
    V sumR = R[o];
#pragma unroll
    for (int i = 1; i < P; ++i) sumR += R[o + i];
    const V G_a = sumR * kg.inv_c;
    V G_u;
    if constexpr (INV) G_u = cum[g] - G_a;
    else G_u = drv[3 + g] - G_a;
    // (skipping the ra and f2 terms behind wave-uniform tests of those coefficients — zero in most default gases — was
    // tried: 8 fewer instructions per member-step and +2.5 % fused fp32 / +3 % fused fp64; the branches cost more than
    // they save, r03/ab_variants.txt)
    V iirf = fma3<V>(kg.ra, G_a, fma3<V>(rr[3 * g + 2], T_old, fma3<V>(rr[3 * g + 1], G_u, rr[3 * g])));
    iirf = fe_min(iirf, km.iirf_max);
    const V alpha = kg.g0 * fe_exp(iirf * kg.inv_g1);
    const V inv_alpha = fe_rcp(alpha);
    // --- step_conc -----------------------------------------------------------------
    V em1[P];
#pragma unroll
    for (int i = 0; i < P; ++i) em1[i] = fe_expm1_neg(kg.ndt_over_tau[i] * inv_alpha);
    V E;
    if constexpr (INV) {
        V num = V(0), den = V(0);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            num += fe_fma(R[o + i], em1[i], R[o + i]);
            den = fe_fma(kg.atc[i], em1[i], den);
        }
        E = (num - (drv[g] - kg.C0)) / (alpha * den);
        cum[g] = fe_fma(E, km.dt, cum[g]);
        out[g] = E;
    } else {
        E = (V)drv[g];
    }
    const V Ea = E * alpha;
    V sumN = (V)S(0);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const V Ri = R[o + i];
        V Rn;
        if constexpr (COMP) {
            const V y = fe_fma(em1[i], fma3<V>(-kg.atc[i], Ea, Ri), Rlo[o + i]);     // the increment plus what earlier sums dropped
            Rn = Ri + y;
            Rlo[o + i] = y - (Rn - Ri);                                            // what THIS sum dropped
        } else {
            Rn = fe_fma(em1[i], fma3<V>(-kg.atc[i], Ea, Ri), Ri);         // R + em1 (R - a tau c E alpha)
        }
        R[o + i] = Rn;
        sumN += Rn;
    }
    const V Cg = kg.C0 + sumN;
    if constexpr (!INV) out[g] = Cg;
    // --- step_forc (terms whose coefficient is zero are skipped: wave-uniform branch) ---
    const auto pos = fe_gt0(Cg);
    V Fg = kg.f2 * (Cg - kg.C0);
    if constexpr (COMP) {
        Fg = kg.f2 * sumN;
        if (kg.f1 != S(0)) {
            const V x = sumN * kg.inv_C0;                                          // C / C0 - 1, to the precision of the excess
            const V u = fe_sel(pos, (V)S(1) + x, (V)S(1));
            const V lg = fe_log(u) + (x - (u - (V)S(1)));                          // log1p(x)
            Fg = fe_sel(pos, fma3<V>(kg.f1, lg, Fg), Fg);
        }
        if (kg.f3 != S(0)) {
            const V den = fe_sqrt(fe_sel(pos, Cg, (V)S(1))) + kg.sqrtC0;
            Fg = fma3<V>(kg.f3, fe_sel(pos, sumN * fe_rcp(den), (V)(-kg.sqrtC0)), Fg);
        }
    } else if constexpr (Lane<V>::W == 1) {
        if (kg.f1 != S(0)) Fg = pos ? fe_fma(kg.f1, fe_log(pos ? Cg * kg.inv_C0 : S(1)), Fg) : Fg;
        if (kg.f3 != S(0)) Fg = fe_fma(kg.f3, (pos ? fe_sqrt(pos ? Cg : S(1)) : S(0)) - kg.sqrtC0, Fg);
    } else {
        if (kg.f1 != S(0)) Fg = fe_sel(pos, fma3<V>(kg.f1, fe_log(fe_sel(pos, Cg * kg.inv_C0, (V)S(1))), Fg), Fg);
        if (kg.f3 != S(0)) Fg = fma3<V>(kg.f3, fe_sel(pos, fe_sqrt(fe_sel(pos, Cg, (V)S(1))), (V)S(0)) - kg.sqrtC0, Fg);
    }
    return Fg;
}

template <typename V, typename L, bool INV, bool COMP>
__device__ __forceinline__ void member_step(const KModel<typename Lane<V>::S>& km, const typename Lane<V>::S* __restrict__ drv,
                                            const V (&rr)[3 * L::G], const V (&qq)[2],
                                            V (&R)[L::SP], V (&S)[2], V (&out)[L::G], V& Tnew, V (&cum)[L::G],
                                            V (&Rlo)[L::SP]) {
    const V T_old = S[0] + S[1];
    V F = (V)drv[6];
    // compiler-only barriers: keep each gas's LDS constant reads inside that gas's code instead of all
    // ~45 being hoisted to the kernel top (VGPR pressure) or out of the fused time loop.  (Issuing gas
    // g+1's reads before gas g's arithmetic was tried: +-1 %, 133 VGPRs; not kept.)
    asm volatile("" ::: "memory");
    F += gas_step<V, L, 0, INV, COMP>(km, km.gas[0], drv, rr, T_old, R, out, cum, Rlo);
    if constexpr (L::G > 1) {
        asm volatile("" ::: "memory");
        F += gas_step<V, L, 1, INV, COMP>(km, km.gas[1], drv, rr, T_old, R, out, cum, Rlo);
    }
    if constexpr (L::G > 2) {
        asm volatile("" ::: "memory");
        F += gas_step<V, L, 2, INV, COMP>(km, km.gas[2], drv, rr, T_old, R, out, cum, Rlo);
    }
    // --- step_temp: S + em1_d (S - q F) ------------------------------------------------
#pragma unroll
    for (int j = 0; j < 2; ++j) S[j] = fma3<V>(km.em1_d[j], fe_fma(-qq[j], F, S[j]), S[j]);
    Tnew = S[0] + S[1];
}
template <typename V, typename L, bool INV = false>
__device__ __forceinline__ void member_step(const KModel<typename Lane<V>::S>& km, const typename Lane<V>::S* __restrict__ drv,
                                            const V (&rr)[3 * L::G], const V (&qq)[2],
                                            V (&R)[L::SP], V (&S)[2], V (&out)[L::G], V& Tnew, V (&cum)[L::G]) {
    V no_Rlo[L::SP];                                     // never touched: COMP = false
    member_step<V, L, INV, false>(km, drv, rr, qq, R, S, out, Tnew, cum, no_Rlo);
}
template <typename V, typename L>
__device__ __forceinline__ void member_step(const KModel<typename Lane<V>::S>& km, const typename Lane<V>::S* __restrict__ drv,
                                            const V (&rr)[3 * L::G], const V (&qq)[2],
                                            V (&R)[L::SP], V (&S)[2], V (&C)[L::G], V& Tnew) {
    V unused[L::G];
    member_step<V, L, false>(km, drv, rr, qq, R, S, C, Tnew, unused);
}

// The shared model is the FIRST kernel argument (by value): its bytes sit at offset 0 of the
// kernarg segment.  With ~45 fp64 constants per 3-gas layout plus the polynomial literals it does
// not fit the 102-SGPR budget (118 SGPR spills -> v_readlane/v_writelane in the VALU stream), so
// each workgroup copies it once into LDS and the lanes read it back with broadcast ds_reads,
// which issue beside the VALU instead of in it.
template <typename T>
__device__ __forceinline__ void stage_model(KModel<T>* dst) {
    constexpr int NW = sizeof(KModel<T>) / sizeof(T);
    const T* src = (const T*)__builtin_amdgcn_kernarg_segment_ptr();
    for (int i = threadIdx.x; i < NW; i += FIVEEQ_BLOCK) reinterpret_cast<T*>(dst)[i] = src[i];
}

template <typename T>
__device__ __forceinline__ void store_stream(T* p, T v) { *p = v; }   // plain store (hfc_conc_kernel only; the step kernels' stored rows go through their NT policy)

// ---------------------------------------------------------------------------------
// Per-wave summary statistics of T for one step: (sum, sum of squares, min, max) over the wave's
// active members, in fp64, written to stats[(wave * n_steps + t) * 4 .. +3] (wave-major, so a
// member sub-range of a larger run addresses its records with a plain pointer offset).
// The 64 lanes are folded in registers with DPP moves (row_shr 1/2/4/8 inside each row of 16
// lanes, then row_bcast15 and row_bcast31 across rows: the gfx9 wave-reduce ladder); lanes with no
// DPP source receive the operation's neutral element.  The total lands in lane 63, which writes
// the 32-byte record.  (A first version used LDS fp64 atomics on one address per wave: 64-way
// serialised, +54 % on the fused kernel; the DPP ladder costs a few hundred cycles per wave-step.)
// One record per wave and step (0.5 B per member-step) replaces the T trajectory when only
// moments are wanted.
// ---------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_f64(const double v, const double neutral) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(neutral), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(neutral), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
struct OpAdd { static __device__ __forceinline__ double f(double a, double b) { return a + b; } };
struct OpMin { static __device__ __forceinline__ double f(double a, double b) { return fmin(a, b); } };
struct OpMax { static __device__ __forceinline__ double f(double a, double b) { return fmax(a, b); } };
template <typename Op>
__device__ __forceinline__ double wave_reduce_to_lane63(double v, const double neutral) {
    v = Op::f(v, dpp_move_f64<0x111, 0xf>(v, neutral));   // row_shr:1
    v = Op::f(v, dpp_move_f64<0x112, 0xf>(v, neutral));   // row_shr:2
    v = Op::f(v, dpp_move_f64<0x114, 0xf>(v, neutral));   // row_shr:4
    v = Op::f(v, dpp_move_f64<0x118, 0xf>(v, neutral));   // row_shr:8   -> lane 15 of each row = row total
    v = Op::f(v, dpp_move_f64<0x142, 0xa>(v, neutral));   // row_bcast:15 into rows 1 and 3
    v = Op::f(v, dpp_move_f64<0x143, 0xc>(v, neutral));   // row_bcast:31 into rows 2 and 3 -> lane 63 = total
    return v;
}

template <typename T>
__device__ __forceinline__ void wave_stats(const bool active, const T Tn, double* __restrict__ out) {
    const double inf = __builtin_inf();
    const double v = (double)Tn;
    const double s1 = wave_reduce_to_lane63<OpAdd>(active ? v : 0.0, 0.0);
    const double s2 = wave_reduce_to_lane63<OpAdd>(active ? v * v : 0.0, 0.0);
    const double mn = wave_reduce_to_lane63<OpMin>(active ? v : inf, inf);
    const double mx = wave_reduce_to_lane63<OpMax>(active ? v : -inf, -inf);
    if ((threadIdx.x & 63) == 63) {
        out[0] = s1;
        out[1] = s2;
        out[2] = mn;
        out[3] = mx;
    }
}

// Packed lanes (two members per lane): the wave covers 128 consecutive members, lanes 0..31 the first 64 and lanes
// 32..63 the second 64, so the ladder stops one step early (no row_bcast:31) and lane 31 / lane 63 write the two
// 64-member records — the record layout [ceil(N/64)][n_steps][4] is the same for every kernel shape.
template <typename Op>
__device__ __forceinline__ double wave_reduce_to_lanes_31_63(double v, const double neutral) {
    v = Op::f(v, dpp_move_f64<0x111, 0xf>(v, neutral));   // row_shr:1
    v = Op::f(v, dpp_move_f64<0x112, 0xf>(v, neutral));   // row_shr:2
    v = Op::f(v, dpp_move_f64<0x114, 0xf>(v, neutral));   // row_shr:4
    v = Op::f(v, dpp_move_f64<0x118, 0xf>(v, neutral));   // row_shr:8
    v = Op::f(v, dpp_move_f64<0x142, 0xa>(v, neutral));   // row_bcast:15 into rows 1 and 3 -> lanes 31, 63 = half totals
    return v;
}
__device__ __forceinline__ void wave_stats(const bool a0, const bool a1, const float2v Tn, double* __restrict__ out_lo,
                                           double* __restrict__ out_hi /* nullptr: the wave has <= 64 members */) {
    const double inf = __builtin_inf();
    const double x = (double)Tn.x, y = (double)Tn.y;
    const double s1 = wave_reduce_to_lanes_31_63<OpAdd>((a0 ? x : 0.0) + (a1 ? y : 0.0), 0.0);
    const double s2 = wave_reduce_to_lanes_31_63<OpAdd>((a0 ? x * x : 0.0) + (a1 ? y * y : 0.0), 0.0);
    const double mn = wave_reduce_to_lanes_31_63<OpMin>(fmin(a0 ? x : inf, a1 ? y : inf), inf);
    const double mx = wave_reduce_to_lanes_31_63<OpMax>(fmax(a0 ? x : -inf, a1 ? y : -inf), -inf);
    const int lane = threadIdx.x & 63;
    double* const out = lane == 31 ? out_lo : (lane == 63 ? out_hi : nullptr);
    if (out != nullptr) {
        out[0] = s1;
        out[1] = s2;
        out[2] = mn;
        out[3] = mx;
    }
}

// Row access of a lane: one element (scalar lanes) or two consecutive elements as ONE 8-byte access (packed lanes; the
// host guarantees even row strides and 8-byte aligned rows before it picks a packed kernel).  `full` = both members of
// a packed lane exist; the last lane of an odd ensemble stores its first member only.
template <typename V>
__device__ __forceinline__ V load_lane(const typename Lane<V>::S* p) { return *reinterpret_cast<const V*>(p); }
__device__ __forceinline__ void store_lane(double* p, double v, bool) { *p = v; }
__device__ __forceinline__ void store_lane(float* p, float v, bool) { *p = v; }
__device__ __forceinline__ void store_lane(float* p, float2v v, bool full) {
    if (full) *reinterpret_cast<float2v*>(p) = v;
    else *p = v.x;
}
// The same with the NON-TEMPORAL policy (NT = true): rows that are read or written once per pass over an ensemble far larger
// than the Infinity Cache, where keeping them resident cannot pay (step_kernel's STREAM form).
template <typename V, bool NT>
__device__ __forceinline__ V load_row(const typename Lane<V>::S* p) {
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const V*>(p));
    else return load_lane<V>(p);
}
template <bool NT, typename S, typename V>
__device__ __forceinline__ void store_row(S* p, V v, bool full) {
    if constexpr (!NT) store_lane(p, v, full);
    else if constexpr (sizeof(V) == sizeof(S)) __builtin_nontemporal_store(v, p);
    else {
        if (full) __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
        else __builtin_nontemporal_store(v.x, p);
    }
}

// ---------------------------------------------------------------------------------
// The time-fused kernel produces one T per lane EVERY step, so it batches the statistics instead
// of running the DPP ladder per step (which costs +20 % fp64 / +70 % fp32 there): each wave parks
// its T values in a wave-private LDS tile [STAT_STEPS][64 (+1 pad)], and every STAT_STEPS steps
// the tile is reduced TRANSPOSED: lane l owns step j = l % 8 and the eighth p = l / 8 of that
// step's 64 members, folds its 8 values serially in fp64, and the 8 partials per step are combined
// with three xor-shuffles (8, 16, 32).  Row stride 65 elements makes both the row writes and the
// strided reads bank-conflict-free (bank = j + 8 p + i mod 32).  A wave's LDS operations complete
// in program order, so only compiler (wavefront-scope) fences are needed, no barrier.
// ---------------------------------------------------------------------------------
constexpr int STAT_STEPS = 8;
constexpr int STAT_ROW = 65;
// min / max as ONE instruction.  fmin()/fmax() on a value the compiler cannot prove canonical get a v_max(x, x) in front
// (sNaN quieting) — 56 of them in the fused kernel's statistics flush; the values here come out of the model's FMAs.  A NaN
// operand is ignored by v_min / v_max like by fmin / fmax (IEEE mode), so the record of a wave with a NaN member is the same.
__device__ __forceinline__ float fe_min_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float fe_max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double fe_min_raw(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double fe_max_raw(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <typename T>
__device__ __forceinline__ void wave_stats_flush(const T* tile /* [STAT_STEPS][STAT_ROW] */, const int count,
                                                 const int n_valid, double* __restrict__ out, const int64_t stride) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const int lane = threadIdx.x & 63;
    const int j = lane & (STAT_STEPS - 1), p = lane >> 3;
    const double inf = __builtin_inf();
    double s1 = 0.0, s2 = 0.0, mn, mx;
    if (n_valid >= 64) {                                   // a full wave (all but the ensemble's last): no per-value tests, and
        T lo_v = tile[j * STAT_ROW + p * 8], hi_v = lo_v;  // min / max in the values' own precision (exact), converted once
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const T t = tile[j * STAT_ROW + p * 8 + i];
            const double v = (double)t;
            s1 += v;
            s2 = __builtin_fma(v, v, s2);
            lo_v = fe_min_raw(lo_v, t);
            hi_v = fe_max_raw(hi_v, t);
        }
        mn = (double)lo_v;
        mx = (double)hi_v;
    } else {
        mn = inf;
        mx = -inf;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = p * 8 + i;
            const double v = (double)tile[j * STAT_ROW + idx];
            if (idx < n_valid) {
                s1 += v;
                s2 = __builtin_fma(v, v, s2);
                mn = fmin(mn, v);
                mx = fmax(mx, v);
            }
        }
    }
#pragma unroll
    for (int sh = 8; sh < 64; sh <<= 1) {
        s1 += __shfl_xor(s1, sh);
        s2 += __shfl_xor(s2, sh);
        mn = fmin(mn, __shfl_xor(mn, sh));
        mx = fmax(mx, __shfl_xor(mx, sh));
    }
    if (p == 0 && j < count) {
        double* o = out + (int64_t)j * stride;
        o[0] = s1;
        o[1] = s2;
        o[2] = mn;
        o[3] = mx;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// Packed lanes: the tile row holds the wave's 64 float2 values (128 members); lane l owns step j = l % 8 and lanes
// 8p .. 8p+7 of it (members 16p .. 16p+15), p = l / 8; p < 4 is the wave's first 64-member record, p >= 4 its second.
__device__ __forceinline__ void wave_stats_flush(const float2v* tile /* [STAT_STEPS][STAT_ROW] */, const int count,
                                                 const int n_valid /* members of this wave, <= 128 */,
                                                 double* __restrict__ out_lo, double* __restrict__ out_hi, const int64_t stride) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const int lane = threadIdx.x & 63;
    const int j = lane & (STAT_STEPS - 1), p = lane >> 3;
    const double inf = __builtin_inf();
    double s1 = 0.0, s2 = 0.0, mn, mx;
    if (n_valid >= 128) {                                  // a full wave: same order of the sums as below, no per-value tests
        const float2v first = tile[j * STAT_ROW + p * 8];
        float lo_v = first.x, hi_v = first.x;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float2v v2 = tile[j * STAT_ROW + p * 8 + i];
            const double x = (double)v2.x, y = (double)v2.y;
            s1 += x;
            s2 = __builtin_fma(x, x, s2);
            s1 += y;
            s2 = __builtin_fma(y, y, s2);
            lo_v = fe_min_raw(fe_min_raw(lo_v, v2.x), v2.y);
            hi_v = fe_max_raw(fe_max_raw(hi_v, v2.x), v2.y);
        }
        mn = (double)lo_v;
        mx = (double)hi_v;
    } else {
        mn = inf;
        mx = -inf;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = p * 8 + i;
            const float2v v2 = tile[j * STAT_ROW + idx];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double v = (double)(c == 0 ? v2.x : v2.y);
                if (2 * idx + c < n_valid) {
                    s1 += v;
                    s2 = __builtin_fma(v, v, s2);
                    mn = fmin(mn, v);
                    mx = fmax(mx, v);
                }
            }
        }
    }
#pragma unroll
    for (int sh = 8; sh < 32; sh <<= 1) {
        s1 += __shfl_xor(s1, sh);
        s2 += __shfl_xor(s2, sh);
        mn = fmin(mn, __shfl_xor(mn, sh));
        mx = fmax(mx, __shfl_xor(mx, sh));
    }
    double* const out = p == 0 ? out_lo : (p == 4 ? out_hi : nullptr);
    if (out != nullptr && j < count) {
        double* o = out + (int64_t)j * stride;
        o[0] = s1;
        o[1] = s2;
        o[2] = mn;
        o[3] = mx;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// ---------------------------------------------------------------------------------
// Kernel 1 — ONE TIMESTEP PER LAUNCH (the north-star form).
// Per member-step HBM traffic (elements): read SP + 2 (state) + 3G + 2 (params),
// write SP + 2 (state) + G + 1 (C, T rows)  ->  A = w (2 SP + 4 G + 7) bytes
// (152 B CO2-only fp64, 248 B for pools 4+1+1 fp64).
// The step's drive record — emissions, cumulative emissions, F_ext and the OUTPUT ROW this step
// is stored at (drive[t][7]; negative = not stored) — is staged through LDS once per workgroup,
// next to the shared model.  One member per lane; a workgroup is ONE wave of 64 members (finer
// dispatch granularity and a trivial barrier: -2 % at 1M members, -3 % at 8M, -7 % at 100k against
// 256-thread workgroups with identical buffers, profiles/r01/ab_variants.txt) and owns the same
// members in every launch.
// CACHE POLICY OF THE ROWS (round 5, profiles/r05/step_row_policy_ab.txt, hbm_rates.txt):
//   * the stored C / T rows are written with the NON-TEMPORAL policy in every form: written once, never read by a stepping
//     kernel, they only displace state and parameter rows from the 256 MiB Infinity Cache that the next step would have hit
//     (-1.4 % per step at 1M fp64 members, -1 % at the 1.25M shard, -3...-8 % on chunk-major runs of 8-25M members);
//   * NT = true (the STREAMED form): state and parameter rows too.  For a launch whose rows cannot survive until the next
//     step anyway — far more members than the cache holds, not scheduled chunk-major — the default policy only adds
//     allocate-and-evict work to every access: -4.5 % per step at 8M fp64 members (0.676 -> 0.707 of 8 TB/s), -9 % at 4M;
//     on a cache-resident ensemble it is the WRONG form (+11...13 % at 1-2M members).  The host picks per call
//     (fiveeq_capi.hip, rows_streamed()).  Same arithmetic: the same bits.
// ---------------------------------------------------------------------------------
// BINS = true: the streamed-histogram form.  Besides everything above, the kernel writes the histogram BIN INDEX of T of every
// step (fiveeq_hist_rows' bin rule, bit for bit; 0xFFFF for a NaN) as one uint16 per member into a ring
// bin_ring[ring_rows][ld] at row t mod ring_rows — 2 bytes per member-step where a ring of T rows takes w — for the
// histogram pass (hist_bins_kernel) to count.  The pass no longer sees T, so the moments stay in the kernel (stats).
constexpr unsigned short BIN_NAN = 0xFFFFu;
// THE BIN RULE — one definition per row precision, used by every kernel that bins a value (the in-loop forms of the step /
// fused kernels, hist_rows_kernel on stored rows, the summary's selection pass), so that "the same counts bit for
// bit" between them is a property of this struct.  (lo, inv_w = n_bins / (hi - lo), n_bins) come in as fp64:
//   fp64 rows:  pos = (v - lo) * inv_w                      in fp64
//   fp32 rows:  pos = fma(v, (float)inv_w, (float)(-lo * inv_w))    in fp32 — one (packed) FMA where the fp64 form cost ~10
//               quarter-rate instructions per lane in kernels whose ceiling is VALU issue (round 4).  Against the fp64 form a
//               member changes bin only within ~2^-23 max(|lo|, |hi|) inv_w of a bin edge (the rounding of scale and offset,
//               in bins): 2^-12 bin for a range that starts near zero (|lo| inv_w ~ n_bins <= 4096, e.g. temperature
//               anomalies), more for a range far from zero in units of its own width (lo = 280, hi = 295, 4096 bins: 0.01 bin)
//               — rows in such absolute units want a range shifted to the anomaly, or fp64 rows
//   bin = pos clamped to [0, n_bins - 1] and truncated; outliers land in the edge bins; a NaN has no bin (BIN_NAN).
// Both forms are monotone in v (rounding is), which the summary's selection relies on: members of a lower bin are <= members
// of a higher one.
// The rule's three constants are plain values (an object with methods made the compiler park it in LDS — promote-alloca — in
// the one-wave step kernel: 768 B of LDS, -1 wave/SIMD, +17 % on the per-step + bins form; measured, profiles/r04/ab_variants.txt).
template <typename S> struct HistRule;
template <> struct HistRule<double> {
    double lo, inv_w, top;
};
template <> struct HistRule<float> {
    float scale, offset, top;
};
__device__ __forceinline__ HistRule<double> make_rule(const double, const double lo, const double inv_w, const int n_bins) {
    return HistRule<double>{lo, inv_w, (double)(n_bins - 1)};
}
__device__ __forceinline__ HistRule<float> make_rule(const float, const double lo, const double inv_w, const int n_bins) {
    // scale and offset are kept FINITE (a range narrower than ~1e-35 would overflow them): pos is then never inf - inf, so a
    // finite or infinite member always clamps into [0, n_bins - 1] and no index can leave the histogram
    const double big = 3.0e38;
    const float scale = (float)fmin(fmax(inv_w, -big), big), offset = (float)fmin(fmax(-lo * inv_w, -big), big);
    return HistRule<float>{scale, offset, (float)(n_bins - 1)};
}
__device__ __forceinline__ unsigned int hist_bin(const HistRule<double> r, const double v) {
    const double pos = (v - r.lo) * r.inv_w;
    const unsigned int b = (unsigned int)(int)fmin(fmax(pos, 0.0), r.top);          // NaN pos -> 0 (fmax / fmin drop the NaN)
    return v == v ? b : (unsigned int)BIN_NAN;
}
__device__ __forceinline__ unsigned int hist_bin_of_pos(const HistRule<float> r, const float pos, const float v) {
    const unsigned int b = (unsigned int)(int)__builtin_amdgcn_fmed3f(pos, 0.0f, r.top);       // v_med3_f32: the clamp in one op
    return v == v ? b : (unsigned int)BIN_NAN;
}
// (rounds 2-3 binned fp32 rows by the fp64 formula — convert, subtract, multiply, clamp, truncate: ~10 quarter-rate instructions
// per lane; the A/B against this one fp32 FMA is profiles/r04/ab_variants.txt, the knob is gone)
__device__ __forceinline__ unsigned int hist_bin(const HistRule<float> r, const float v) {
    return hist_bin_of_pos(r, __builtin_fmaf(v, r.scale, r.offset), v);
}
// two members of a packed lane: one v_pk_fma_f32; returns bin(v.x) | bin(v.y) << 16
__device__ __forceinline__ unsigned int hist_bin2(const HistRule<float> r, const float2v v) {
    const float2v pos = __builtin_elementwise_fma(v, (float2v)r.scale, (float2v)r.offset);
    return hist_bin_of_pos(r, pos.x, v.x) | (hist_bin_of_pos(r, pos.y, v.y) << 16);
}

#ifdef FIVEEQ_STEP_WAVES
#define FIVEEQ_STEP_ATTR __attribute__((amdgpu_waves_per_eu(FIVEEQ_STEP_WAVES, FIVEEQ_STEP_WAVES)))
#else
#define FIVEEQ_STEP_ATTR
#endif
template <typename V, int P0, int P1, int P2, bool BINS = false, bool NT = false>
__global__ __launch_bounds__(FIVEEQ_STEP_BLOCK) FIVEEQ_STEP_ATTR void step_kernel(
    const KModel<typename Lane<V>::S> km, const typename Lane<V>::S* __restrict__ drive, const int n_steps, const int t,
    const int64_t n, const int64_t ld,
    const typename Lane<V>::S* __restrict__ r, const typename Lane<V>::S* __restrict__ q,
    typename Lane<V>::S* __restrict__ R, typename Lane<V>::S* __restrict__ S,
    typename Lane<V>::S* __restrict__ C_traj /* [n_rows][G][ld] or nullptr */,
    typename Lane<V>::S* __restrict__ T_traj /* [n_rows][ld] or nullptr */,
    const int n_rows, double* __restrict__ stats /* [ceil(n/64)][n_steps][4] or nullptr */,
    unsigned short* __restrict__ bin_ring /* BINS: [ring_rows][ld], row t mod ring_rows */, const int ring_rows,
    const double hist_lo, const double hist_inv_w, const int n_bins) {
    using L = Layout<P0, P1, P2>;
    using T = typename Lane<V>::S;
    constexpr int W = Lane<V>::W;                 // members per lane
    constexpr bool NTT = true;                    // the stored C / T rows: written once, never read by a stepping kernel
    __shared__ T drv[DRIVE_STRIDE];
    const int64_t m = ((int64_t)blockIdx.x * FIVEEQ_STEP_BLOCK + threadIdx.x) * W;     // this lane's first member
    const bool active = m < n;
    const bool full = m + (W - 1) < n;            // every member of the lane exists
    // idle tail lanes load a valid (aligned) member and store nothing
    const int64_t mm = active ? m : ((n - 1) & ~(int64_t)(W - 1));
    // Issue order matters for the workgroup's critical path: first the (tiny) shared loads, then
    // all 19 row loads, and only then the LDS writes + barrier, so the staging round trip is
    // overlapped with the row round trip instead of preceding it (+1.3 % at 1M members, neutral
    // elsewhere: profiles/r01/ab_variants.txt).
    __shared__ KModel<T> km_s;
    constexpr int NW = sizeof(KModel<T>) / sizeof(T);
    static_assert(NW <= FIVEEQ_STEP_BLOCK, "model must stage in one pass");
    const T* kargs = (const T*)__builtin_amdgcn_kernarg_segment_ptr();
    T stage_v = T(0), drv_v = T(0);
    if (threadIdx.x < NW) stage_v = kargs[threadIdx.x];
    if (threadIdx.x < DRIVE_STRIDE) drv_v = drive[(int64_t)t * DRIVE_STRIDE + threadIdx.x];
    const KModel<T>& kmr = km_s;

    V rr[3 * L::G], qq[2], Rv[L::SP], Sv[2], Cv[L::G];
#pragma unroll
    for (int k = 0; k < L::SP; ++k) Rv[k] = load_row<V, NT>(R + k * ld + mm);
#pragma unroll
    for (int k = 0; k < 2; ++k) Sv[k] = load_row<V, NT>(S + k * ld + mm);
#pragma unroll
    for (int k = 0; k < 3 * L::G; ++k) rr[k] = load_row<V, NT>(r + k * ld + mm);
#pragma unroll
    for (int k = 0; k < 2; ++k) qq[k] = load_row<V, NT>(q + k * ld + mm);

    if (threadIdx.x < NW) reinterpret_cast<T*>(&km_s)[threadIdx.x] = stage_v;
    if (threadIdx.x < DRIVE_STRIDE) drv[threadIdx.x] = drv_v;
    __syncthreads();

    V Tn = (V)T(0);
    {
        member_step<V, L>(kmr, drv, rr, qq, Rv, Sv, Cv, Tn);
        if (active) {
#pragma unroll
        for (int k = 0; k < L::SP; ++k) store_row<NT>(R + k * ld + m, Rv[k], full);
#pragma unroll
        for (int k = 0; k < 2; ++k) store_row<NT>(S + k * ld + m, Sv[k], full);
        const int row = __builtin_amdgcn_readfirstlane((int)drv[7]);     // wave-uniform: scalar test + offsets
        if (row >= 0 && row < n_rows) {
            if (C_traj != nullptr) {
                T* c = C_traj + (int64_t)row * L::G * ld + m;
#pragma unroll
                for (int g = 0; g < L::G; ++g) store_row<NTT>(c + g * ld, Cv[g], full);
            }
            if (T_traj != nullptr) store_row<NTT>(T_traj + (int64_t)row * ld + m, Tn, full);
        }
        if constexpr (BINS) {                                            // the histogram bin of T, 2 bytes per member
            unsigned short* o = bin_ring + (int64_t)(t % ring_rows) * ld + m;
            const HistRule<T> rule = make_rule(T(0), hist_lo, hist_inv_w, n_bins);
            if constexpr (W == 1) {
                *o = (unsigned short)hist_bin(rule, Tn);
            } else {
                const unsigned int b01 = hist_bin2(rule, Tn);
                if (full) *reinterpret_cast<unsigned int*>(o) = b01;
                else *o = (unsigned short)(b01 & 0xffffu);
            }
        }
        }
    }
    if (stats != nullptr) {
        const int64_t n_rec = (n + 63) >> 6;                             // one record per 64 members
        const int64_t wave = (int64_t)blockIdx.x * (FIVEEQ_STEP_BLOCK / 64) + (threadIdx.x >> 6);
        if constexpr (W == 1) {
            if (wave < n_rec) wave_stats(active, Tn, stats + (wave * n_steps + t) * 4);
        } else {
            if (2 * wave < n_rec)
                wave_stats(active, full, Tn, stats + (2 * wave * n_steps + t) * 4,
                           2 * wave + 1 < n_rec ? stats + ((2 * wave + 1) * n_steps + t) * 4 : nullptr);
        }
    }
}

// ---------------------------------------------------------------------------------
// Kernel 2 — TIME-FUSED: one launch advances [t_begin, t_end); a member's state and
// parameters stay in registers for the whole span, the drive table is staged into LDS
// FIVEEQ_FUSED_CHUNK steps at a time, and only the C/T rows of stored steps go to HBM.
// Per member-step traffic: w (G + 1) + w (2 SP + 3 G + 6) / n_steps.
// Same member_step() as kernel 1: results are bit-identical.
// ---------------------------------------------------------------------------------
//
// INV = true: concentration-driven form.  drive[t][0..2] are target concentrations, cumE [G][ld] is
// per-member cumulative-emission state (in/out), and C_traj receives the DIAGNOSED EMISSIONS.
// (112 VGPRs at fp64 4+1+1 = 4 waves/SIMD; launch-bounds hints for 5 or 6 waves spill: -3 % / -16 %.)
template <typename V, int P0, int P1, int P2, bool INV, bool BINS = false, bool COMP = false>
__global__ __launch_bounds__(FIVEEQ_BLOCK) void fused_kernel(
    const KModel<typename Lane<V>::S> km, const typename Lane<V>::S* __restrict__ drive, const int n_steps,
    const int t_begin, const int t_end, const int64_t n, const int64_t ld,
    const typename Lane<V>::S* __restrict__ r, const typename Lane<V>::S* __restrict__ q,
    typename Lane<V>::S* __restrict__ R, typename Lane<V>::S* __restrict__ S,
    typename Lane<V>::S* __restrict__ cumE /* [G][ld], INV only */,
    typename Lane<V>::S* __restrict__ C_traj /* [n_rows][G][ld] or nullptr */,
    typename Lane<V>::S* __restrict__ T_traj /* [n_rows][ld] or nullptr */,
    const int n_rows, double* __restrict__ stats /* [ceil(n/64)][n_steps][4] or nullptr */,
    unsigned short* __restrict__ bin_ring /* BINS: [ring_rows][ld] */, const int ring_rows, const double hist_lo,
    const double hist_inv_w, const int n_bins) {
    using L = Layout<P0, P1, P2>;
    using T = typename Lane<V>::S;
    constexpr int W = Lane<V>::W;                 // members per lane
    static_assert(!(INV && BINS), "no streamed histograms in the concentration-driven form");
    __shared__ T drv[FIVEEQ_FUSED_CHUNK * DRIVE_STRIDE];
    __shared__ V stat_tile[FIVEEQ_BLOCK / 64][STAT_STEPS * STAT_ROW];
    __shared__ KModel<T> km_s;
    stage_model(&km_s);
    const KModel<T>& kmr = km_s;

    const int64_t m = ((int64_t)blockIdx.x * FIVEEQ_BLOCK + threadIdx.x) * W;       // this lane's first member
    const bool active = m < n;
    const bool full = m + (W - 1) < n;
    const int64_t mm = active ? m : 0;    // idle tail lanes shadow member 0 and store nothing
    const int64_t n_rec = (n + 63) >> 6;                                             // statistics records: one per 64 members
    const int64_t wave = (int64_t)blockIdx.x * (FIVEEQ_BLOCK / 64) + (threadIdx.x >> 6);
    const bool wave_live = stats != nullptr && wave * W < n_rec;
    V* const tile = stat_tile[threadIdx.x >> 6];
    const int n_valid = (int)min((int64_t)64 * W, n - wave * 64 * W);               // members of this wave (<= 0: none)
    int ks = 0;                                                                      // steps parked in the tile
    const HistRule<T> rule = make_rule(T(0), hist_lo, hist_inv_w, n_bins);           // (BINS only)

    V rr[3 * L::G], qq[2], Rv[L::SP], Sv[2], Cv[L::G], Tn, cum[L::G];
    V Rlo[L::SP];                                         // COMP: the compensation words (zero at launch: they do not cross HBM)
    if constexpr (COMP) {
#pragma unroll
        for (int k = 0; k < L::SP; ++k) Rlo[k] = (V)T(0);
    }
    if constexpr (INV) {
#pragma unroll
        for (int g = 0; g < L::G; ++g) cum[g] = cumE[g * ld + mm];
    }
#pragma unroll
    for (int k = 0; k < L::SP; ++k) Rv[k] = load_lane<V>(R + k * ld + mm);
#pragma unroll
    for (int k = 0; k < 2; ++k) Sv[k] = load_lane<V>(S + k * ld + mm);
#pragma unroll
    for (int k = 0; k < 3 * L::G; ++k) rr[k] = load_lane<V>(r + k * ld + mm);
#pragma unroll
    for (int k = 0; k < 2; ++k) qq[k] = load_lane<V>(q + k * ld + mm);

    for (int tc = t_begin; tc < t_end; tc += FIVEEQ_FUSED_CHUNK) {
        const int nt = min(FIVEEQ_FUSED_CHUNK, t_end - tc);
        __syncthreads();                                  // previous chunk fully consumed
        for (int i = threadIdx.x; i < nt * DRIVE_STRIDE; i += FIVEEQ_BLOCK)
            drv[i] = drive[(int64_t)tc * DRIVE_STRIDE + i];
        __syncthreads();
        for (int k = 0; k < nt; ++k) {
            const T* d = &drv[k * DRIVE_STRIDE];
            member_step<V, L, INV, COMP>(kmr, d, rr, qq, Rv, Sv, Cv, Tn, cum, Rlo);
            // the output row is wave-uniform: read it once into an SGPR so that the row test is a
            // scalar branch and the row offsets are scalar arithmetic, not 64-bit VALU per lane
            const int row = __builtin_amdgcn_readfirstlane((int)d[7]);
            if (row >= 0 && row < n_rows) {
                if (active) {
                    if (C_traj != nullptr) {
                        T* c = C_traj + (int64_t)row * L::G * ld + m;
#pragma unroll
                        for (int g = 0; g < L::G; ++g) store_lane(c + g * ld, Cv[g], full);
                    }
                    if (T_traj != nullptr) store_lane(T_traj + (int64_t)row * ld + m, Tn, full);
                }
            }
            if constexpr (BINS) {
                if (active) {
                    unsigned short* o = bin_ring + (int64_t)((tc + k) % ring_rows) * ld + m;     // scalar row offset
                    if constexpr (W == 1) {
                        *o = (unsigned short)hist_bin(rule, Tn);
                    } else {
                        const unsigned int b01 = hist_bin2(rule, Tn);
                        if (full) *reinterpret_cast<unsigned int*>(o) = b01;                // both members: one 4-byte store
                        else *o = (unsigned short)(b01 & 0xffffu);
                    }
                }
            }
            if (wave_live) {
                tile[ks * STAT_ROW + (threadIdx.x & 63)] = Tn;
                if (++ks == STAT_STEPS || tc + k + 1 == t_end) {
                    const int64_t t_first = (int64_t)(tc + k + 1 - ks);
                    if constexpr (W == 1) {
                        wave_stats_flush(tile, ks, n_valid, stats + (wave * n_steps + t_first) * 4, 4);
                    } else {
                        wave_stats_flush(tile, ks, n_valid, stats + (2 * wave * n_steps + t_first) * 4,
                                         2 * wave + 1 < n_rec ? stats + ((2 * wave + 1) * n_steps + t_first) * 4 : nullptr, 4);
                    }
                    ks = 0;
                }
            }
        }
    }
    if (active) {
#pragma unroll
        for (int k = 0; k < L::SP; ++k) store_lane(R + k * ld + m, Rv[k], full);
#pragma unroll
        for (int k = 0; k < 2; ++k) store_lane(S + k * ld + m, Sv[k], full);
        if constexpr (INV) {
#pragma unroll
            for (int g = 0; g < L::G; ++g) cumE[g * ld + m] = cum[g];
        }
    }
}

// ---------------------------------------------------------------------------------
// Kernel 2c — SMALL ENSEMBLES: the time-fused step with ONE MEMBER SPREAD OVER A QUAD OF LANES (round 5).
//
// A 10k-member ensemble (BASELINE configs[1]) is 157 waves for 1024 SIMDs: every wave is alone on its SIMD, and a lone
// wave issues one vector instruction per ~3.7-4.2 ns whatever the instruction and however independent its neighbours are
// (tools/microbench/valu_rates.hip, "waves/SIMD 1": 9-10 nominal cycles for v_fma_f64 and for v_mov_b32 alike).  What such
// a run costs is therefore the NUMBER OF INSTRUCTIONS ONE WAVE ISSUES PER STEP — not bytes, not FLOPs, not occupancy — and
// the way to shorten it is to hand parts of a member's step to lanes that would otherwise not exist:
//   * LPM = 4 (layouts with a 4-pool gas and nothing else: CO2-only): lane 4m + i carries POOL i of member m.  Its expm1,
//     its pool update and its slice of the state are the lane's own (one expm1 chain per wave-step instead of four); the
//     alpha closure, the forcing and the thermal boxes are computed by all four lanes alike (redundant lanes are free:
//     the instruction is issued once per wave either way).  The two sums over pools are folded with quad_perm DPP moves
//     in the per-step kernel's order ((R0 + R1) + R2) + R3, every lane of the quad computing the same sum from the same
//     four values: the bits do not change.  4x the waves of the one-member-per-lane form, 16 members per wave;
//   * the shared model is read from the KERNEL ARGUMENT (scalar loads, hoisted out of the time loop) instead of being
//     re-read from LDS every step: with one wave per SIMD the registers are there (512 VGPRs), and an LDS round trip that
//     nothing hides is ~100 cycles of the wave's time;
//   * the step's drive record is read one step AHEAD (LDS, broadcast), so that its latency lies under the previous step.
// LPM = 1 is the same kernel without the spreading (any single-gas layout): what the register-resident constants buy alone.
// Per-wave statistics as in the fused kernel (the same records, bit for bit); no histogram ring: those runs take the fused
// kernel.  Same arithmetic, operation for operation, as member_step(): bit-identical results (tested against the per-step path).
// ---------------------------------------------------------------------------------
template <int K>
__device__ __forceinline__ double quad_bcast(const double v) {           // lane 4q + K of every quad, to the whole quad
    // (mov_dpp, not update_dpp: every lane has a source, so there is no "old" value to initialise — 16 v_mov less per step)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), K * 0x55, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), K * 0x55, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ float quad_bcast(const float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), K * 0x55, 0xf, 0xf, true));
}

template <typename T, int P0, int LPM, bool STATS>
__global__ __launch_bounds__(FIVEEQ_SMALL_BLOCK) __attribute__((amdgpu_waves_per_eu(1, 2))) void small_kernel(
    const KModel<T> km, const T* __restrict__ drive, const int n_steps, const int t_begin, const int t_end,
    const int64_t n, const int64_t ld, const T* __restrict__ r, const T* __restrict__ q, T* __restrict__ R,
    T* __restrict__ S, T* __restrict__ C_traj /* [n_rows][1][ld] or nullptr */, T* __restrict__ T_traj /* [n_rows][ld] or nullptr */,
    const int n_rows, double* __restrict__ stats /* [ceil(n/64)][n_steps][4] or nullptr */) {
    static_assert(LPM == 1 || (LPM == 4 && P0 == 4), "a quad of lanes carries the four pools of one gas");
    static_assert(LPM == 1 || FIVEEQ_SMALL_BLOCK == 256, "quad form: one workgroup = 64 members = one statistics record");
    constexpr int MPB = FIVEEQ_SMALL_BLOCK / LPM;                        // members per workgroup
    __shared__ T drv[FIVEEQ_FUSED_CHUNK * DRIVE_STRIDE];
    __shared__ int row_s[FIVEEQ_FUSED_CHUNK];                            // the steps' output rows, converted once per chunk
    // per-64-member statistics records, batched over STAT_STEPS steps and folded by wave_stats_flush() exactly like the fused
    // kernel's (same tile layout, same order: the same record bits).  One lane per member: a tile per wave.  A quad per member:
    // the workgroup's four waves hold 16 members each = ONE record; they share a tile and wave 0 folds it between two barriers.
    // A compile-time variant (STATS): as a run-time test in the time loop it cost the statistics-free run 10 % (0.415 -> 0.456 us
    // per step at 10k members).
    __shared__ T stat_tile[!STATS ? 1 : (LPM == 1 ? FIVEEQ_SMALL_BLOCK / 64 : 1)][!STATS ? 1 : STAT_STEPS * STAT_ROW];
    const int64_t rec = LPM == 1 ? (int64_t)blockIdx.x * (FIVEEQ_SMALL_BLOCK / 64) + (threadIdx.x >> 6) : (int64_t)blockIdx.x;
    const bool rec_live = STATS && rec < ((n + 63) >> 6);                 // uniform over the wave (LPM = 1) / the workgroup (LPM = 4)
    const int n_valid = (int)min((int64_t)64, n - rec * 64);
    T* const tile = STATS ? stat_tile[LPM == 1 ? threadIdx.x >> 6 : 0] : nullptr;
    int ks = 0;
    const int lane = threadIdx.x;
    const int sub = lane % LPM;                                          // the pool this lane carries (LPM = 4)
    const int64_t m = (int64_t)blockIdx.x * MPB + lane / LPM;
    const bool active = m < n;
    const int64_t mm = active ? m : 0;                                   // idle tail lanes shadow member 0 and store nothing
    const KGas<T>& kg = km.gas[0];                                       // kernel argument: scalar loads, loop-invariant

    T rr[3], qq[2], Sv[2], Rv[LPM == 1 ? P0 : 1];
    T ndt[LPM == 1 ? P0 : 1], natc[LPM == 1 ? P0 : 1];                  // -dt / tau_i and -(a_i tau_i c) of this lane's pool(s)
    if constexpr (LPM == 1) {
#pragma unroll
        for (int i = 0; i < P0; ++i) Rv[i] = R[i * ld + mm], ndt[i] = kg.ndt_over_tau[i], natc[i] = -kg.atc[i];
    } else {
        Rv[0] = R[sub * ld + mm];
        ndt[0] = sub == 0 ? kg.ndt_over_tau[0] : (sub == 1 ? kg.ndt_over_tau[1] : (sub == 2 ? kg.ndt_over_tau[2] : kg.ndt_over_tau[3]));
        natc[0] = -(sub == 0 ? kg.atc[0] : (sub == 1 ? kg.atc[1] : (sub == 2 ? kg.atc[2] : kg.atc[3])));
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) Sv[k] = S[k * ld + mm];
#pragma unroll
    for (int k = 0; k < 3; ++k) rr[k] = r[k * ld + mm];
#pragma unroll
    for (int k = 0; k < 2; ++k) qq[k] = q[k * ld + mm];
    // LPM = 4: ONE store per step and lane — lane 0 of the quad writes C, lane 1 writes T (its own base pointer; null = this
    // lane stores nothing)
    T* const out_q = !active ? nullptr : (sub == 0 ? (C_traj ? C_traj + m : nullptr) : (sub == 1 ? (T_traj ? T_traj + m : nullptr) : nullptr));

    // the sum over pools of the CURRENT state, in the per-step kernel's order ((R0 + R1) + R2) + R3.  Computed here once; each
    // step's own sum over the NEW pools, (((0 + R0') + R1') + R2') + R3', is then the next step's: the same additions on the
    // same values (0 + x = x), except that a zero may come out with the other sign, which alpha = g0 exp(iIRF / g1) — the
    // sum's only consumer — cannot see.
    T sumR;
    if constexpr (LPM == 1) {
        sumR = Rv[0];
#pragma unroll
        for (int i = 1; i < P0; ++i) sumR += Rv[i];
    } else {
        sumR = quad_bcast<0>(Rv[0]);
        sumR += quad_bcast<1>(Rv[0]);
        sumR += quad_bcast<2>(Rv[0]);
        sumR += quad_bcast<3>(Rv[0]);
    }

    for (int tc = t_begin; tc < t_end; tc += FIVEEQ_FUSED_CHUNK) {
        const int nt = min(FIVEEQ_FUSED_CHUNK, t_end - tc);
        __syncthreads();                                                 // (one wave: the previous chunk is consumed)
        for (int i = threadIdx.x; i < nt * DRIVE_STRIDE; i += FIVEEQ_SMALL_BLOCK) {
            const T v = drive[(int64_t)tc * DRIVE_STRIDE + i];
            drv[i] = v;
            if ((i & (DRIVE_STRIDE - 1)) == 7) row_s[i >> 3] = (int)v;
        }
        __syncthreads();
        T E = drv[0], cumE = drv[3], Fx = drv[6];                        // step tc
        int rowv = row_s[0];
        for (int k = 0; k < nt; ++k) {
            const int kn = k + 1 < nt ? k + 1 : k;                       // the NEXT step's record, asked for now
            const T En = drv[kn * DRIVE_STRIDE], cumEn = drv[kn * DRIVE_STRIDE + 3], Fxn = drv[kn * DRIVE_STRIDE + 6];
            const int rowvn = row_s[kn];
            // ---- member_step(), operation for operation (gas_step<.., g = 0, INV = false>) ----
            const T T_old = Sv[0] + Sv[1];
            const T G_a = sumR * kg.inv_c;
            const T G_u = cumE - G_a;
            T iirf = fe_fma(kg.ra, G_a, fe_fma(rr[2], T_old, fe_fma(rr[1], G_u, rr[0])));
            iirf = fe_min(iirf, km.iirf_max);
            const T alpha = kg.g0 * fe_exp(iirf * kg.inv_g1);
            const T inv_alpha = fe_rcp(alpha);
            const T Ea = E * alpha;
            T sumN = T(0);
            if constexpr (LPM == 1) {
                T em1[P0];
#pragma unroll
                for (int i = 0; i < P0; ++i) em1[i] = fe_expm1_neg(ndt[i] * inv_alpha);
#pragma unroll
                for (int i = 0; i < P0; ++i) {
                    const T Rn = fe_fma(em1[i], fe_fma(natc[i], Ea, Rv[i]), Rv[i]);
                    Rv[i] = Rn;
                    sumN += Rn;
                }
            } else {
                const T em1 = fe_expm1_neg(ndt[0] * inv_alpha);
                const T Rn = fe_fma(em1, fe_fma(natc[0], Ea, Rv[0]), Rv[0]);
                Rv[0] = Rn;
                sumN += quad_bcast<0>(Rn);
                sumN += quad_bcast<1>(Rn);
                sumN += quad_bcast<2>(Rn);
                sumN += quad_bcast<3>(Rn);
            }
            sumR = sumN;
            const T Cg = kg.C0 + sumN;
            const bool pos = Cg > T(0);
            T Fg = kg.f2 * (Cg - kg.C0);
            if (kg.f1 != T(0)) {                                         // gas_step()'s values, selected instead of branched around:
                const T lg = fe_log(pos ? Cg * kg.inv_C0 : T(1));        // straight-line code schedules better in a lone wave (-1...-3 %)
                const T with_log = fe_fma(kg.f1, lg, Fg);
                Fg = pos ? with_log : Fg;
            }
            if (kg.f3 != T(0)) {
                const T sq = fe_sqrt(pos ? Cg : T(1));
                Fg = fe_fma(kg.f3, (pos ? sq : T(0)) - kg.sqrtC0, Fg);
            }
            T F = Fx;
            F += Fg;
#pragma unroll
            for (int j = 0; j < 2; ++j) Sv[j] = fe_fma(km.em1_d[j], fe_fma(-qq[j], F, Sv[j]), Sv[j]);
            const T Tn = Sv[0] + Sv[1];
            // ---- the step's stored rows ----
            const int row = __builtin_amdgcn_readfirstlane(rowv);
            if (row >= 0 && row < n_rows) {
                if constexpr (LPM == 1) {
                    if (active) {
                        if (C_traj != nullptr) C_traj[(int64_t)row * ld + m] = Cg;
                        if (T_traj != nullptr) T_traj[(int64_t)row * ld + m] = Tn;
                    }
                } else {
                    if (out_q != nullptr) out_q[(int64_t)row * ld] = sub == 0 ? Cg : Tn;
                }
            }
            if constexpr (STATS) if (rec_live) {
                if (LPM == 1 || sub == 0) tile[ks * STAT_ROW + (LPM == 1 ? (threadIdx.x & 63) : (threadIdx.x >> 2))] = Tn;
                if (++ks == STAT_STEPS || tc + k + 1 == t_end) {
                    double* const out = stats + (rec * n_steps + (tc + k + 1 - ks)) * 4;
                    if constexpr (LPM == 1) {
                        wave_stats_flush(tile, ks, n_valid, out, 4);
                    } else {
                        __syncthreads();
                        if (threadIdx.x < 64) wave_stats_flush(tile, ks, n_valid, out, 4);
                        __syncthreads();
                    }
                    ks = 0;
                }
            }
            E = En, cumE = cumEn, Fx = Fxn, rowv = rowvn;
        }
    }
    if (active) {
        if constexpr (LPM == 1) {
#pragma unroll
            for (int i = 0; i < P0; ++i) R[i * ld + m] = Rv[i];
#pragma unroll
            for (int k = 0; k < 2; ++k) S[k * ld + m] = Sv[k];
        } else {
            R[sub * ld + m] = Rv[0];
            if (sub < 2) S[sub * ld + m] = sub == 0 ? Sv[0] : Sv[1];
        }
    }
}

// The same for layouts with SEVERAL gases, one member per lane: member_step() itself on a model that lives in registers
// (scalar loads of the kernel argument, hoisted out of the time loop: with at most two waves per SIMD the ~45 constants of
// three gases fit beside the state) and on a drive record held in registers and read one step ahead.  What a launch-bound
// multi-gas ensemble gains over the fused kernel is the LDS round trips per step that nothing hides when a wave is alone on
// its SIMD.  (A quad per gas would carry 4 members per wave: worth it below ~4k members only; not built.)
// COMP = true (fp32 only): the compensated form of gas_step (a compensation word per pool in registers, the forcing from the excess
// C - C0) on this kernel — what a launch-bound fp32 ensemble takes under EnsembleEngine(compensated=True); every layout, the
// single-gas ones included (fiveeq_run_small_comp_f32).
template <typename T, int P0, int P1, int P2, bool STATS, bool COMP = false>
__global__ __launch_bounds__(FIVEEQ_SMALL_BLOCK) __attribute__((amdgpu_waves_per_eu(1, 2))) void small_multi_kernel(
    const KModel<T> km, const T* __restrict__ drive, const int n_steps, const int t_begin, const int t_end,
    const int64_t n, const int64_t ld, const T* __restrict__ r, const T* __restrict__ q, T* __restrict__ R,
    T* __restrict__ S, T* __restrict__ C_traj /* [n_rows][G][ld] or nullptr */, T* __restrict__ T_traj /* [n_rows][ld] or nullptr */,
    const int n_rows, double* __restrict__ stats /* [ceil(n/64)][n_steps][4] or nullptr */) {
    using L = Layout<P0, P1, P2>;
    __shared__ T drv[FIVEEQ_FUSED_CHUNK * DRIVE_STRIDE];
    __shared__ int row_s[FIVEEQ_FUSED_CHUNK];
    __shared__ T stat_tile[!STATS ? 1 : FIVEEQ_SMALL_BLOCK / 64][!STATS ? 1 : STAT_STEPS * STAT_ROW];     // as in small_kernel, one lane per member
    const int64_t rec = (int64_t)blockIdx.x * (FIVEEQ_SMALL_BLOCK / 64) + (threadIdx.x >> 6);
    const bool rec_live = STATS && rec < ((n + 63) >> 6);
    const int n_valid = (int)min((int64_t)64, n - rec * 64);
    T* const tile = STATS ? stat_tile[threadIdx.x >> 6] : nullptr;
    int ks = 0;
    // fp64: the model, word by word, into VECTOR registers.  Left to itself the compiler keeps the ~45 constants of three
    // gases in scalar registers, runs out of them (two each) and spills — 83 v_readlane per step, 1.25 us per step instead of
    // 0.94 at 10k members (r05/ab_variants.txt section 4).  fp32 constants fit the scalar file and stay there (0.51 against 0.61).
    KModel<T> kl;
    {
        constexpr int NW = sizeof(KModel<T>) / sizeof(T);
        const T* src = reinterpret_cast<const T*>(&km);
        T* dst = reinterpret_cast<T*>(&kl);
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            T v = src[i];
            if constexpr (sizeof(T) == 8) asm("" : "+v"(v));             // (not volatile: words of absent gases are dropped)
            dst[i] = v;
        }
    }
    const int64_t m = (int64_t)blockIdx.x * FIVEEQ_SMALL_BLOCK + threadIdx.x;
    const bool active = m < n;
    const int64_t mm = active ? m : 0;
    T rr[3 * L::G], qq[2], Rv[L::SP], Sv[2], Cv[L::G], Tn;
    T Rlo[L::SP], no_cum[L::G];                                          // COMP: the compensation words (zero at launch)
    if constexpr (COMP) {
#pragma unroll
        for (int k = 0; k < L::SP; ++k) Rlo[k] = T(0);
    }
#pragma unroll
    for (int k = 0; k < L::SP; ++k) Rv[k] = R[k * ld + mm];
#pragma unroll
    for (int k = 0; k < 2; ++k) Sv[k] = S[k * ld + mm];
#pragma unroll
    for (int k = 0; k < 3 * L::G; ++k) rr[k] = r[k * ld + mm];
#pragma unroll
    for (int k = 0; k < 2; ++k) qq[k] = q[k * ld + mm];
    for (int tc = t_begin; tc < t_end; tc += FIVEEQ_FUSED_CHUNK) {
        const int nt = min(FIVEEQ_FUSED_CHUNK, t_end - tc);
        __syncthreads();
        for (int i = threadIdx.x; i < nt * DRIVE_STRIDE; i += FIVEEQ_SMALL_BLOCK) {
            const T v = drive[(int64_t)tc * DRIVE_STRIDE + i];
            drv[i] = v;
            if ((i & (DRIVE_STRIDE - 1)) == 7) row_s[i >> 3] = (int)v;
        }
        __syncthreads();
        T cur[DRIVE_STRIDE - 1];
#pragma unroll
        for (int j = 0; j < DRIVE_STRIDE - 1; ++j) cur[j] = drv[j];
        int rowv = row_s[0];
        for (int k = 0; k < nt; ++k) {
            const int kn = k + 1 < nt ? k + 1 : k;                       // the NEXT step's record, asked for now
            T nxt[DRIVE_STRIDE - 1];
#pragma unroll
            for (int j = 0; j < DRIVE_STRIDE - 1; ++j) nxt[j] = drv[kn * DRIVE_STRIDE + j];
            const int rowvn = row_s[kn];
            member_step<T, L, false, COMP>(kl, cur, rr, qq, Rv, Sv, Cv, Tn, no_cum, Rlo);
            const int row = __builtin_amdgcn_readfirstlane(rowv);
            if (row >= 0 && row < n_rows && active) {
                if (C_traj != nullptr) {
                    T* c = C_traj + (int64_t)row * L::G * ld + m;
#pragma unroll
                    for (int g = 0; g < L::G; ++g) c[g * ld] = Cv[g];
                }
                if (T_traj != nullptr) T_traj[(int64_t)row * ld + m] = Tn;
            }
            if constexpr (STATS) if (rec_live) {
                tile[ks * STAT_ROW + (threadIdx.x & 63)] = Tn;
                if (++ks == STAT_STEPS || tc + k + 1 == t_end) {
                    wave_stats_flush(tile, ks, n_valid, stats + (rec * n_steps + (tc + k + 1 - ks)) * 4, 4);
                    ks = 0;
                }
            }
#pragma unroll
            for (int j = 0; j < DRIVE_STRIDE - 1; ++j) cur[j] = nxt[j];
            rowv = rowvn;
        }
    }
    if (active) {
#pragma unroll
        for (int k = 0; k < L::SP; ++k) R[k * ld + m] = Rv[k];
#pragma unroll
        for (int k = 0; k < 2; ++k) S[k * ld + m] = Sv[k];
    }
}

// ---------------------------------------------------------------------------------
// Kernel 2d — SMALL MULTI-GAS ENSEMBLES: one member per OCTET of lanes (round 6; layout 4 + 1 + 1, the default three-gas set).
//
// The quad idea of small_kernel carried to three gases: lanes 0-3 of an octet hold the four pools of gas 0, lane 4 the pool of
// gas 1, lane 5 the pool of gas 2 (lanes 6, 7 shadow lane 5 and store nothing).  Every lane runs ONE alpha closure, ONE expm1
// chain, ONE pool update and ONE forcing — its own gas's, with that gas's constants selected into registers once — where the
// one-member-per-lane form (small_multi_kernel) runs three closures, six expm1 chains and three forcings per wave-step: a third
// of the instructions per wave, on 8x the waves.  What crosses lanes, all of it DPP moves inside a row of 16 lanes:
//   * gas 0's sum over pools, folded inside its quad in member_step()'s order (((0 + R0) + R1) + R2) + R3; a single-pool gas's
//     sum is 0 + R, lane-local; one select between the two;
//   * the three forcings: quad_perm broadcasts lane 0 / lane 1 of every quad (quad 0: F_0, F_0; quad 1: F_1, F_2), row_shr:4 /
//     row_shl:4 under a bank mask carry them into the other quad, and every lane adds F_ext + F_0 + F_1 + F_2 in that order.
// The thermal boxes are computed by all eight lanes alike.  Same operations on the same values in the same order as
// member_step(): bit-identical results (tested against the per-step path, fp64 and fp32).  No per-wave statistics (a record
// is 64 members = eight of these waves; runs with collect_stats take the one-lane form).
// ---------------------------------------------------------------------------------
template <int CTRL, int BANKS>
__device__ __forceinline__ double dpp_merge(const double old, const double src) {       // lanes of the banks in BANKS: src moved by CTRL; others: old
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xf, BANKS, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xf, BANKS, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int BANKS>
__device__ __forceinline__ float dpp_merge(const float old, const float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, BANKS, false));
}
constexpr int DPP_ROW_SHL4 = 0x104, DPP_ROW_SHR4 = 0x114;          // lane i reads lane i + 4 / lane i - 4 of its row of 16

template <typename T>
__global__ __launch_bounds__(FIVEEQ_SMALL_BLOCK) __attribute__((amdgpu_waves_per_eu(1, 2))) void small_octet_kernel(
    const KModel<T> km, const T* __restrict__ drive, const int n_steps, const int t_begin, const int t_end,
    const int64_t n, const int64_t ld, const T* __restrict__ r, const T* __restrict__ q, T* __restrict__ R,
    T* __restrict__ S, T* __restrict__ C_traj /* [n_rows][3][ld] or nullptr */, T* __restrict__ T_traj /* [n_rows][ld] or nullptr */,
    const int n_rows) {
    constexpr int MPB = FIVEEQ_SMALL_BLOCK / 8;                          // members per workgroup
    __shared__ T drv[FIVEEQ_FUSED_CHUNK * DRIVE_STRIDE];
    __shared__ int row_s[FIVEEQ_FUSED_CHUNK];
    const int lane = threadIdx.x;
    const int o = lane & 7;                                              // position in the octet
    const int g = o < 4 ? 0 : (o == 4 ? 1 : 2);                          // this lane's gas
    const int prow = o < 4 ? o : (o == 4 ? 4 : 5);                       // ... and its pool's row of R
    const bool co2 = o < 4;
    const int64_t m = (int64_t)blockIdx.x * MPB + lane / 8;
    const bool active = m < n;
    const int64_t mm = active ? m : 0;                                   // idle tail lanes shadow member 0 and store nothing
    // this lane's gas, selected once from the kernel argument (scalar loads) into vector registers
#define FIVEEQ_PICK(field) (g == 0 ? km.gas[0].field : (g == 1 ? km.gas[1].field : km.gas[2].field))
    const T ndt = co2 ? (o == 0 ? km.gas[0].ndt_over_tau[0] : (o == 1 ? km.gas[0].ndt_over_tau[1] : (o == 2 ? km.gas[0].ndt_over_tau[2] : km.gas[0].ndt_over_tau[3])))
                      : (g == 1 ? km.gas[1].ndt_over_tau[0] : km.gas[2].ndt_over_tau[0]);
    const T natc = -(co2 ? (o == 0 ? km.gas[0].atc[0] : (o == 1 ? km.gas[0].atc[1] : (o == 2 ? km.gas[0].atc[2] : km.gas[0].atc[3])))
                         : (g == 1 ? km.gas[1].atc[0] : km.gas[2].atc[0]));
    const T g0 = FIVEEQ_PICK(g0), inv_g1 = FIVEEQ_PICK(inv_g1), ra = FIVEEQ_PICK(ra), inv_c = FIVEEQ_PICK(inv_c);
    const T C0 = FIVEEQ_PICK(C0), inv_C0 = FIVEEQ_PICK(inv_C0), sqrtC0 = FIVEEQ_PICK(sqrtC0);
    const T f1 = FIVEEQ_PICK(f1), f2 = FIVEEQ_PICK(f2), f3 = FIVEEQ_PICK(f3);
#undef FIVEEQ_PICK
    const bool has_log = f1 != T(0), has_sqrt = f3 != T(0);
    T rr[3], qq[2], Sv[2];
    T Rv = R[prow * ld + mm];
#pragma unroll
    for (int k = 0; k < 2; ++k) Sv[k] = S[k * ld + mm];
#pragma unroll
    for (int k = 0; k < 3; ++k) rr[k] = r[(3 * g + k) * ld + mm];
#pragma unroll
    for (int k = 0; k < 2; ++k) qq[k] = q[k * ld + mm];
    // ONE store per step and lane: lanes 0 / 4 / 5 write their gas's C, lane 1 writes T (null = this lane stores nothing)
    T* out_p = nullptr;
    int64_t out_stride = 0;
    if (active) {
        if (o == 1) out_p = T_traj ? T_traj + m : nullptr, out_stride = ld;
        else if (o == 0 || o == 4 || o == 5) out_p = C_traj ? C_traj + g * ld + m : nullptr, out_stride = 3 * ld;
    }
    // the sum over this lane's gas's pools of the CURRENT state, as small_kernel keeps it (a step's own sum is the next step's)
    T sumR;
    {
        T s4 = quad_bcast<0>(Rv);
        s4 += quad_bcast<1>(Rv);
        s4 += quad_bcast<2>(Rv);
        s4 += quad_bcast<3>(Rv);
        sumR = co2 ? s4 : Rv;
    }
    for (int tc = t_begin; tc < t_end; tc += FIVEEQ_FUSED_CHUNK) {
        const int nt = min(FIVEEQ_FUSED_CHUNK, t_end - tc);
        __syncthreads();
        for (int i = threadIdx.x; i < nt * DRIVE_STRIDE; i += FIVEEQ_SMALL_BLOCK) {
            const T v = drive[(int64_t)tc * DRIVE_STRIDE + i];
            drv[i] = v;
            if ((i & (DRIVE_STRIDE - 1)) == 7) row_s[i >> 3] = (int)v;
        }
        __syncthreads();
        T E = drv[g], cumE = drv[3 + g], Fx = drv[6];                    // step tc: this lane's gas's emission and cumulative emission
        int rowv = row_s[0];
        for (int k = 0; k < nt; ++k) {
            const int kn = k + 1 < nt ? k + 1 : k;                       // the NEXT step's record, asked for now
            const T En = drv[kn * DRIVE_STRIDE + g], cumEn = drv[kn * DRIVE_STRIDE + 3 + g], Fxn = drv[kn * DRIVE_STRIDE + 6];
            const int rowvn = row_s[kn];
            // ---- gas_step<.., g, INV = false>, operation for operation, for THIS lane's gas and pool ----
            const T T_old = Sv[0] + Sv[1];
            const T G_a = sumR * inv_c;
            const T G_u = cumE - G_a;
            T iirf = fe_fma(ra, G_a, fe_fma(rr[2], T_old, fe_fma(rr[1], G_u, rr[0])));
            iirf = fe_min(iirf, km.iirf_max);
            const T alpha = g0 * fe_exp(iirf * inv_g1);
            const T inv_alpha = fe_rcp(alpha);
            const T Ea = E * alpha;
            const T em1 = fe_expm1_neg(ndt * inv_alpha);
            const T Rn = fe_fma(em1, fe_fma(natc, Ea, Rv), Rv);
            Rv = Rn;
            T s4 = T(0);
            s4 += quad_bcast<0>(Rn);
            s4 += quad_bcast<1>(Rn);
            s4 += quad_bcast<2>(Rn);
            s4 += quad_bcast<3>(Rn);
            const T s1 = T(0) + Rn;
            const T sumN = co2 ? s4 : s1;
            sumR = sumN;
            const T Cg = C0 + sumN;
            const bool pos = Cg > T(0);
            T Fg = f2 * (Cg - C0);
            {
                const T lg = fe_log(pos ? Cg * inv_C0 : T(1));
                const T with_log = fe_fma(f1, lg, Fg);
                Fg = (has_log && pos) ? with_log : Fg;
                const T sq = fe_sqrt(pos ? Cg : T(1));
                const T with_sqrt = fe_fma(f3, (pos ? sq : T(0)) - sqrtC0, Fg);
                Fg = has_sqrt ? with_sqrt : Fg;
            }
            // ---- the three gases' forcings to every lane of the octet ----
            const T a0 = quad_bcast<0>(Fg);                              // quad 0: F_0 (lane 0's); quad 1: F_1 (lane 4's)
            const T a1 = quad_bcast<1>(Fg);                              // quad 0: F_0 (lane 1's); quad 1: F_2 (lane 5's)
            const T F0 = dpp_merge<DPP_ROW_SHR4, 0xA>(a0, a0);           // quads 1, 3 of the row take their left neighbour's
            const T F1 = dpp_merge<DPP_ROW_SHL4, 0x5>(a0, a0);           // quads 0, 2 take their right neighbour's
            const T F2 = dpp_merge<DPP_ROW_SHL4, 0x5>(a1, a1);
            T F = Fx;
            F += F0;
            F += F1;
            F += F2;
#pragma unroll
            for (int j = 0; j < 2; ++j) Sv[j] = fe_fma(km.em1_d[j], fe_fma(-qq[j], F, Sv[j]), Sv[j]);
            const T Tn = Sv[0] + Sv[1];
            const int row = __builtin_amdgcn_readfirstlane(rowv);
            if (row >= 0 && row < n_rows) {
                if (out_p != nullptr) out_p[(int64_t)row * out_stride] = o == 1 ? Tn : Cg;
            }
            E = En, cumE = cumEn, Fx = Fxn, rowv = rowvn;
        }
    }
    if (active) {
        if (o < 6) R[prow * ld + m] = Rv;
        if (o < 2) S[o * ld + m] = o == 0 ? Sv[0] : Sv[1];
    }
}

// ---------------------------------------------------------------------------------
// LDS counter increment with ONE round of wave-level aggregation (the histogram passes and the pick pass).  In the first decades of a run every member's T sits in a
// handful of bins: 64 lanes adding to the same LDS dword serialise (the first two 64-step chunks of a streamed run took
// 1.9 and 0.8 ms in the histogram pass against 0.35 ms later).
// So: the wave looks at the counter of its first lane; if at least 16 lanes want that same counter, ONE of them adds their
// number and the others of the group add nothing; every other lane adds as usual.  `key` identifies the counter (the bin;
// ~0u = this lane has nothing to count), `p` / `inc` are where and what this lane would add.  All lanes of the wave that
// are active at the call site must call it (it is a wave-level operation on the active lanes).
__device__ __forceinline__ void wave_lds_add(unsigned int* p, const unsigned int inc, const unsigned int key) {
    const unsigned int k0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)key);
    const unsigned long long same = __ballot(key == k0);
    const int n_same = __popcll(same);
    if (n_same >= 16) {                                               // wave-uniform
        if (key == k0) {
            if (k0 != ~0u && (int)(threadIdx.x & 63) == __ffsll((long long)same) - 1) atomicAdd(p, inc * (unsigned int)n_same);
        } else if (key != ~0u) {
            atomicAdd(p, inc);
        }
    } else if (key != ~0u) {
        atomicAdd(p, inc);
    }
}

// ---------------------------------------------------------------------------------
// Kernel 5 — shard-computable Latin hypercube.  u[k][m - m0] for members m0 <= m < m0 + n of a
// design over n_total members: u = (pi_k(m) + jitter_k(m)) / n_total, pi_k a KEYED BIJECTION of
// [0, n_total) (4-round balanced Feistel network on the next even power of two, cycle-walked back
// into range), jitter a 24-bit counter-based hash placed mid-cell, so u lies strictly inside
// stratum pi_k(m).  Pure function of (seed, dimension, m, n_total): any rank computes exactly its
// own members, on its own device, and the design is the same for every world size.  Integer
// arithmetic + one fp64 add that is EXACT for n_total <= 2^28 (28 stratum bits + 25 jitter bits <= 53; the C ABI
// refuses larger designs) + one correctly rounded division: params.lhs_rows (NumPy) reproduces it bit for bit.
// ---------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t lhs_mix64(uint64_t z) {      // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t lhs_dim_key(uint64_t seed, int dim) {
    return lhs_mix64(seed + 0x9e3779b97f4a7c15ULL * (uint64_t)(dim + 1));
}
__host__ __device__ __forceinline__ uint64_t lhs_permute(uint64_t m, uint64_t n_total, int half_bits, uint64_t key) {
    const uint64_t mask = (1ULL << half_bits) - 1ULL;
    uint64_t x = m;
    do {
        uint64_t left = x >> half_bits, right = x & mask;
#pragma unroll
        for (int rnd = 0; rnd < 4; ++rnd) {
            const uint64_t f = lhs_mix64(right ^ (key + 0xd1342543de82ef95ULL * (uint64_t)(rnd + 1))) & mask;
            const uint64_t nl = right;
            right = left ^ f;
            left = nl;
        }
        x = (left << half_bits) | right;
    } while (x >= n_total);                                    // cycle-walk: the domain is < 4 n_total
    return x;
}
__global__ __launch_bounds__(FIVEEQ_BLOCK) void lhs_kernel(const uint64_t seed, const int64_t n_total, const int half_bits,
                                                           const int64_t m0, const int64_t n, const int dim0,
                                                           const int64_t ld, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * FIVEEQ_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int dim = dim0 + (int)blockIdx.y;
    const uint64_t key = lhs_dim_key(seed, dim);
    const uint64_t m = (uint64_t)(m0 + i);
    const uint64_t stratum = lhs_permute(m, (uint64_t)n_total, half_bits, key);
    const uint64_t jbits = lhs_mix64(m ^ (key * 0xff51afd7ed558ccdULL + 0xc4ceb9fe1a85ec53ULL)) >> 40;   // 24 bits
    const double jitter = ((double)jbits + 0.5) * 0x1.0p-24;                                               // (0, 1)
    out[(int64_t)blockIdx.y * ld + i] = ((double)stratum + jitter) / (double)n_total;
}

// ---------------------------------------------------------------------------------
// Kernel 3 — ensemble form of the reference's calculate_hfc_conc
// (U_FaIR/concentrations.py:5: emissions[0]*exp(-time)): out[k][m] = e0[m]*exp(-time[k]).
// exp(-time[k]) is shared by every member: each workgroup evaluates a tile of 256 time
// points once into LDS, then every lane scales its member by the staged factors.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(FIVEEQ_BLOCK) void hfc_conc_kernel(
    const int64_t n, const int64_t ld, const int n_time,
    const double* __restrict__ e0, const double* __restrict__ time, double* __restrict__ out) {
    __shared__ double decay[FIVEEQ_BLOCK];
    const int64_t m = (int64_t)blockIdx.x * FIVEEQ_BLOCK + threadIdx.x;
    const bool active = m < n;
    const double e = active ? e0[m] : 0.0;
    for (int k0 = 0; k0 < n_time; k0 += FIVEEQ_BLOCK) {
        const int nk = min(FIVEEQ_BLOCK, n_time - k0);
        __syncthreads();
        if ((int)threadIdx.x < nk) decay[threadIdx.x] = exp(-time[k0 + threadIdx.x]);
        __syncthreads();
        if (active)
            for (int k = 0; k < nk; ++k) store_stream(&out[(int64_t)(k0 + k) * ld + m], e * decay[k]);
    }
}

// ---------------------------------------------------------------------------------
// Kernel 4 — fixed-bin histograms of stored rows (all-timestep percentiles).
// hist[row][bin] += #members with lo + bin*w <= x < lo + (bin+1)*w ; values outside [lo, hi) go to
// the edge bins, NaNs are skipped.  One workgroup = one row x one chunk of members: privatised LDS
// histogram (ds_add_u32), then only the non-zero bins are added to the global 64-bit counters.  The
// host sizes the chunk so that the grid still fills the chip (>= ~2048 workgroups) but no finer: the
// global atomics of the flush, not the read, were the cost at 16384 members per chunk (60 us per
// 12.5M-member row in round 1).  Reads each stored value once: 8 (4) B per member and row.
// ---------------------------------------------------------------------------------
constexpr int HIST_CHUNK_MIN = 16384;
constexpr int HIST_MAX_BINS = 4096;
// (Tried in round 4 and not kept: 2 or 4 SUB-HISTOGRAMS per workgroup, lane l counting into number l mod n, bank-shifted, to
// spare the LDS atomic unit same-address collisions.  The passes got SLOWER — 5.4 -> 7.6 -> 13.4 us per 12.5M-member row of
// bin indices — because the larger LDS footprint halves / quarters the resident waves: these passes run at the box's plain
// copy rate for their access width and are bound by memory-level parallelism, not by LDS atomics.  profiles/r04/ab_variants.txt)

template <typename T>
__global__ __launch_bounds__(FIVEEQ_BLOCK) void hist_rows_kernel(const int64_t n, const int64_t ld, const int64_t chunk,
                                                                 const T* __restrict__ rows, const double lo_all,
                                                                 const double inv_w_all, const int n_bins,
                                                                 unsigned long long* __restrict__ hist,
                                                                 const double* __restrict__ ranges /* [rows][2] or nullptr */) {
    __shared__ unsigned int h[HIST_MAX_BINS];
    for (int b = threadIdx.x; b < n_bins; b += FIVEEQ_BLOCK) h[b] = 0u;
    __syncthreads();
    const int64_t row = blockIdx.y;
    // ranges != nullptr: every row has its own (lo, hi) in device memory (the summary pass: each row's global extrema);
    // a row with hi <= lo is constant and lands in bin 0
    const double lo = ranges ? ranges[row * 2] : lo_all;
    const double inv_w = ranges ? (ranges[row * 2 + 1] > lo ? (double)n_bins / (ranges[row * 2 + 1] - lo) : 0.0) : inv_w_all;
    const int64_t m0 = (int64_t)blockIdx.x * chunk;
    const int64_t m1 = min(m0 + chunk, n);
    const T* x = rows + row * ld;
    const HistRule<T> rule = make_rule(T(0), lo, inv_w, n_bins);
    auto count = [&](const T xv) {
        const unsigned int b = hist_bin(rule, xv);                                    // a NaN has no bin and is not counted
        const bool ok = b != (unsigned int)BIN_NAN;
        wave_lds_add(&h[ok ? b : 0u], 1u, ok ? b : ~0u);
    };
    // the same value counted with a PLAIN LDS atomic: for rows that do not crowd into a few bins (see hist_bins_kernel: the
    // crowding test of wave_lds_add runs once per group of four loads, on the first of them)
    auto count_plain = [&](const T xv) {
        const unsigned int b = hist_bin(rule, xv);
        if (b != (unsigned int)BIN_NAN) atomicAdd(&h[b], 1u);
    };
    int64_t m = m0 + threadIdx.x;
    for (; m + 3 * FIVEEQ_BLOCK < m1; m += 4 * FIVEEQ_BLOCK) {      // four independent loads in flight per lane
        const T v0 = x[m], v1 = x[m + FIVEEQ_BLOCK], v2 = x[m + 2 * FIVEEQ_BLOCK], v3 = x[m + 3 * FIVEEQ_BLOCK];
        const unsigned int b0 = hist_bin(rule, v0);
        const unsigned int k0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)b0);
        if (__popcll(__ballot(b0 == k0)) >= 16) {                    // wave-uniform: a crowded row
            count(v0);
            count(v1);
            count(v2);
            count(v3);
        } else {
            count_plain(v0);
            count_plain(v1);
            count_plain(v2);
            count_plain(v3);
        }
    }
    for (; m < m1; m += FIVEEQ_BLOCK) count(x[m]);
    __syncthreads();
    unsigned long long* out = hist + row * n_bins;
    for (int b = threadIdx.x; b < n_bins; b += FIVEEQ_BLOCK) {
        const unsigned int c = h[b];
        if (c) atomicAdd(&out[b], (unsigned long long)c);
    }
}

// ---------------------------------------------------------------------------------
// Kernel 4b — the pass of the bin-index ring: hist[row][b] += #members whose stored bin index is b (BIN_NAN skipped).
// Same grid shape and LDS privatisation as hist_rows_kernel; reads 2 bytes per member and row, four members per 8-byte load.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(FIVEEQ_BLOCK) void hist_bins_kernel(const int64_t n, const int64_t ld, const int64_t chunk,
                                                                 const unsigned short* __restrict__ rows, const int n_bins,
                                                                 unsigned long long* __restrict__ hist) {
    __shared__ unsigned int h[HIST_MAX_BINS];
    for (int b = threadIdx.x; b < n_bins; b += FIVEEQ_BLOCK) h[b] = 0u;
    __syncthreads();
    const int64_t row = blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * chunk;          // chunk is a multiple of 8 * FIVEEQ_BLOCK (host)
    const int64_t m1 = min(m0 + chunk, n);
    const unsigned short* x = rows + row * ld;
    auto count = [&](const unsigned int b) {
        const bool ok = b < (unsigned int)n_bins;
        wave_lds_add(&h[ok ? b : 0u], 1u, ok ? b : ~0u);
    };
    // rows 16-byte aligned: 8 members per lane and load (1 KiB per wave-instruction: the box copies 17 % faster at 16 than at
    // 8 bytes per lane), two loads in flight, over the whole strides of 8 x 256 members; what is left of the chunk (fewer
    // than 2048 members, only in the row's last chunk) goes one member per lane
    const bool wide = ((((uintptr_t)x) | ((uintptr_t)(ld * 2))) & 15) == 0;
    int64_t m = m0 + (int64_t)threadIdx.x * 4;
    if (wide) {
        // Eight members per lane.  The wave-level aggregation of wave_lds_add exists for rows whose members crowd into a
        // handful of bins (the first decades of a run); it costs ~12 instructions per member, and beside a VALU-bound fused
        // kernel this pass is paid in ISSUE SLOTS, not in bandwidth.  So the crowding test runs ONCE per load, on the lane's
        // first member: a crowded row takes the aggregated path for all eight, every other row plain LDS atomics.
        auto count8 = [&](const uint4 v) {
            const unsigned int b0 = v.x & 0xffffu;
            const unsigned int k0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)b0);
            if (__popcll(__ballot(b0 == k0)) >= 16) {                   // wave-uniform
                count(b0);
                count(v.x >> 16);
                count(v.y & 0xffffu);
                count(v.y >> 16);
                count(v.z & 0xffffu);
                count(v.z >> 16);
                count(v.w & 0xffffu);
                count(v.w >> 16);
            } else {
                auto plain = [&](const unsigned int b) {
                    if (b < (unsigned int)n_bins) atomicAdd(&h[b], 1u);
                };
                plain(b0);
                plain(v.x >> 16);
                plain(v.y & 0xffffu);
                plain(v.y >> 16);
                plain(v.z & 0xffffu);
                plain(v.z >> 16);
                plain(v.w & 0xffffu);
                plain(v.w >> 16);
            }
        };
        constexpr int64_t STRIDE = 8 * FIVEEQ_BLOCK;
        const int64_t whole = m0 + (m1 - m0) / STRIDE * STRIDE;
        int64_t m8 = m0 + (int64_t)threadIdx.x * 8;
        for (; m8 + STRIDE < whole; m8 += 2 * STRIDE) {
            const uint4 v = *reinterpret_cast<const uint4*>(x + m8);
            const uint4 u = *reinterpret_cast<const uint4*>(x + m8 + STRIDE);
            count8(v);
            count8(u);
        }
        if (m8 < whole) count8(*reinterpret_cast<const uint4*>(x + m8));
        for (int64_t r = whole + threadIdx.x; r < m1; r += FIVEEQ_BLOCK) count(x[r]);
        m = m1;
    }
    for (; m < m1; m += 4 * FIVEEQ_BLOCK)                     // unaligned rows, and the ragged tail of the last chunk
        for (int j = 0; j < 4 && m + j < m1; ++j) count(x[m + j]);
    __syncthreads();
    unsigned long long* out = hist + row * n_bins;
    for (int b = threadIdx.x; b < n_bins; b += FIVEEQ_BLOCK) {
        const unsigned int c = h[b];
        if (c) atomicAdd(&out[b], (unsigned long long)c);
    }
}

// ---------------------------------------------------------------------------------
// Kernels 6a-6c — the END-OF-RUN SUMMARY as HIP passes (form (i): exact percentiles by selection).
// The host side (fiveeqscm_amd/distributed.py) needs, per output row of T over this rank's members:
//   6a  the moments (sum, sum of squares, min, max)                      row_moments_kernel + row_moments_fold_kernel
//   --  a 4096-bin histogram between the GLOBAL min and max              hist_rows_kernel with per-row ranges (RANGED)
//   6c  the members of the histogram bins that hold the wanted order statistics, compacted                     select_bins_kernel
//   6d  the order statistics, picked out of those candidates by radix selection                                select_pick_kernel
// Each pass reads the rows once with 16-byte loads; nothing else of the ensemble's size moves.
// ---------------------------------------------------------------------------------
template <typename T> struct Wide;                       // 16 bytes of row per lane and load
template <> struct Wide<double> { using V = double2; static constexpr int N = 2; };
template <> struct Wide<float> { using V = float4; static constexpr int N = 4; };
__device__ __forceinline__ double wide_get(const double2& v, int i) { return i == 0 ? v.x : v.y; }
__device__ __forceinline__ float wide_get(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// 6a.  partial[row][chunk][4] = (sum, sum of squares, min, max) of members [chunk * `chunk`, ...) of the row, fp64 sums
// of the exactly converted elements; min / max ignore NaNs (like the kernels' own wave records), the sums propagate them.
// Fixed summation order (lane-strided, xor-shuffle tree, wave order): the same bits on every run.
template <typename T>
__global__ __launch_bounds__(FIVEEQ_BLOCK) void row_moments_kernel(const int64_t n, const int64_t ld, const int64_t chunk,
                                                                   const T* __restrict__ rows, double* __restrict__ partial) {
    using WV = typename Wide<T>::V;
    constexpr int WN = Wide<T>::N;
    __shared__ double red[FIVEEQ_BLOCK / 64][4];
    const int64_t row = blockIdx.y;
    const int64_t m0 = (int64_t)blockIdx.x * chunk;          // chunk is a multiple of WN * FIVEEQ_BLOCK (host)
    const int64_t m1 = min(m0 + chunk, n);
    const T* x = rows + row * ld;
    const double inf = __builtin_inf();
    double s1 = 0.0, s2 = 0.0, mn = inf, mx = -inf;
    auto take = [&](const T xv) {
        const double v = (double)xv;
        s1 += v;
        s2 = __builtin_fma(v, v, s2);
        mn = fmin(mn, v);
        mx = fmax(mx, v);
    };
    const bool wide = ((((uintptr_t)x) | ((uintptr_t)(ld * sizeof(T)))) & 15) == 0;
    int64_t m = m0 + (int64_t)threadIdx.x * WN;
    if (wide) {
        for (; m + 2 * WN * FIVEEQ_BLOCK + WN - 1 < m1; m += 3 * WN * FIVEEQ_BLOCK) {     // three independent loads in flight
            const WV a = *reinterpret_cast<const WV*>(x + m);
            const WV b = *reinterpret_cast<const WV*>(x + m + WN * FIVEEQ_BLOCK);
            const WV c = *reinterpret_cast<const WV*>(x + m + 2 * WN * FIVEEQ_BLOCK);
#pragma unroll
            for (int j = 0; j < WN; ++j) take(wide_get(a, j));
#pragma unroll
            for (int j = 0; j < WN; ++j) take(wide_get(b, j));
#pragma unroll
            for (int j = 0; j < WN; ++j) take(wide_get(c, j));
        }
        for (; m + WN - 1 < m1; m += WN * FIVEEQ_BLOCK) {
            const WV a = *reinterpret_cast<const WV*>(x + m);
#pragma unroll
            for (int j = 0; j < WN; ++j) take(wide_get(a, j));
        }
    }
    for (; m < m1; m += WN * FIVEEQ_BLOCK)                      // unaligned rows, and the ragged tail of the last chunk
        for (int j = 0; j < WN && m + j < m1; ++j) take(x[m + j]);
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
        s1 += __shfl_xor(s1, sh);
        s2 += __shfl_xor(s2, sh);
        mn = fmin(mn, __shfl_xor(mn, sh));
        mx = fmax(mx, __shfl_xor(mx, sh));
    }
    if ((threadIdx.x & 63) == 0) {
        double* r = red[threadIdx.x >> 6];
        r[0] = s1, r[1] = s2, r[2] = mn, r[3] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = red[0][0], b = red[0][1], c = red[0][2], d = red[0][3];
#pragma unroll
        for (int w = 1; w < FIVEEQ_BLOCK / 64; ++w) {
            a += red[w][0];
            b += red[w][1];
            c = fmin(c, red[w][2]);
            d = fmax(d, red[w][3]);
        }
        double* o = partial + (row * gridDim.x + blockIdx.x) * 4;
        o[0] = a, o[1] = b, o[2] = c, o[3] = d;
    }
}
// moments[row][4] = the partials of a row folded in a fixed order (one wave per row)
__global__ __launch_bounds__(64) void row_moments_fold_kernel(const int64_t chunks, const double* __restrict__ partial,
                                                              double* __restrict__ moments) {
    const int64_t row = blockIdx.x;
    const double inf = __builtin_inf();
    double s1 = 0.0, s2 = 0.0, mn = inf, mx = -inf;
    for (int64_t c = threadIdx.x; c < chunks; c += 64) {
        const double* p = partial + (row * chunks + c) * 4;
        s1 += p[0];
        s2 += p[1];
        mn = fmin(mn, p[2]);
        mx = fmax(mx, p[3]);
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
        s1 += __shfl_xor(s1, sh);
        s2 += __shfl_xor(s2, sh);
        mn = fmin(mn, __shfl_xor(mn, sh));
        mx = fmax(mx, __shfl_xor(mx, sh));
    }
    if (threadIdx.x == 0) {
        double* o = moments + row * 4;
        o[0] = s1, o[1] = s2, o[2] = mn, o[3] = mx;
    }
}

// 6c.  SELECTION.  The histogram of pass 2 says, exactly, how many members lie in each bin, and the bin rule is monotone in
// the value: the order statistic of global index i lies in the bin b with cdf[b-1] <= i < cdf[b], and is the
// (i - cdf[b-1])-th smallest member OF THAT BIN.  So the host marks the bins that hold wanted order statistics
// (binmask[row]: one bit per bin) and this pass, computing every member's bin with the SAME rule from the same (lo, hi),
// appends the members of marked bins — the CANDIDATES, a few thousandths of the row — to cand[row][...] (any order).
// Compaction: the wave's candidates take consecutive places in a workgroup LDS buffer (one LDS atomic per wave-load that has
// any), which is appended to the row's global buffer with ONE global atomic per workgroup; a workgroup whose buffer is full
// (a row with heavy ties) appends its further candidates directly.  cand_n[row] counts every candidate, stored or not; the
// host sizes cap from the histogram, so it never overflows unless the caller passed a smaller one.
constexpr int SELECT_LDS_CAND = 2048;
template <typename T>
__global__ __launch_bounds__(FIVEEQ_BLOCK) void select_bins_kernel(const int64_t n, const int64_t ld, const int64_t chunk,
                                                                   const T* __restrict__ rows, const double* __restrict__ ranges,
                                                                   const int n_bins, const unsigned int* __restrict__ binmask,
                                                                   T* __restrict__ cand, const int64_t cap,
                                                                   unsigned long long* __restrict__ cand_n) {
    using WV = typename Wide<T>::V;
    constexpr int WN = Wide<T>::N;
    __shared__ unsigned int mask_s[HIST_MAX_BINS / 32];
    __shared__ T buf[SELECT_LDS_CAND];
    __shared__ unsigned int buf_next, buf_valid;
    __shared__ unsigned long long g_base;
    const int64_t row = blockIdx.y;
    const int mask_words = (n_bins + 31) >> 5;
    if ((int)threadIdx.x < mask_words) mask_s[threadIdx.x] = binmask[row * mask_words + threadIdx.x];
    if (threadIdx.x == 0) {
        buf_next = 0u;
        buf_valid = 0xffffffffu;
    }
    __syncthreads();
    const double lo = ranges[row * 2], hi = ranges[row * 2 + 1];
    const HistRule<T> rule = make_rule(T(0), lo, hi > lo ? (double)n_bins / (hi - lo) : 0.0, n_bins);      // pass 2's rule for this row
    const int64_t m0 = (int64_t)blockIdx.x * chunk;             // chunk is a multiple of WN * FIVEEQ_BLOCK (host)
    const int64_t m1 = min(m0 + chunk, n);
    const T* x = rows + row * ld;
    T* const out = cand + row * cap;
    const int lane = threadIdx.x & 63;

    // one value per lane: place it if its bin is marked.  `have` = this lane holds a member.
    auto take = [&](const T v, const bool have) {
        const unsigned int b = hist_bin(rule, v);
        const bool is_c = have && b != (unsigned int)BIN_NAN && ((mask_s[b >> 5] >> (b & 31u)) & 1u);
        const unsigned long long cm = __ballot(is_c);
        if (cm != 0ull) {                                        // wave-uniform; a few per cent of the wave-loads of a smooth row
            const unsigned int total = (unsigned int)__popcll(cm);
            const unsigned int rank = (unsigned int)__popcll(cm & ((1ull << lane) - 1ull));
            unsigned int pos = 0u;
            if (lane == 0) pos = atomicAdd(&buf_next, total);
            pos = (unsigned int)__builtin_amdgcn_readfirstlane((int)pos);
            if (pos + total <= (unsigned int)SELECT_LDS_CAND) {
                if (is_c) buf[pos + rank] = v;
            } else {                                             // the workgroup's buffer is full: straight to the row's buffer
                if (lane == 0) atomicMin(&buf_valid, pos);
                unsigned long long gp = 0ull;
                if (lane == 0) gp = atomicAdd(&cand_n[row], (unsigned long long)total);
                gp = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(gp >> 32)) << 32) |
                     (unsigned int)__builtin_amdgcn_readfirstlane((int)(gp & 0xffffffffull));
                if (is_c && (int64_t)(gp + rank) < cap) out[gp + rank] = v;
            }
        }
    };
    const bool wide = ((((uintptr_t)x) | ((uintptr_t)(ld * sizeof(T)))) & 15) == 0;
    // every lane of a wave runs the same number of iterations (take() is a wave-level operation): the loop bound is on the
    // wave's first member, lanes past the end of the chunk carry have = false
    const int64_t wave_m = m0 + (int64_t)(threadIdx.x & ~63) * WN;
    int64_t m = m0 + (int64_t)threadIdx.x * WN;
    constexpr int64_t STEP = (int64_t)WN * FIVEEQ_BLOCK;
    int64_t wm = wave_m;
    if (wide) {
        for (; wm + STEP + 64 * WN <= m1; wm += 2 * STEP, m += 2 * STEP) {      // two independent 16-byte loads in flight
            const WV a = *reinterpret_cast<const WV*>(x + m);
            const WV b = *reinterpret_cast<const WV*>(x + m + STEP);
#pragma unroll
            for (int j = 0; j < WN; ++j) take(wide_get(a, j), true);
#pragma unroll
            for (int j = 0; j < WN; ++j) take(wide_get(b, j), true);
        }
    }
    for (; wm < m1; wm += STEP, m += STEP) {
        if (wide && wm + 64 * WN <= m1) {
            const WV a = *reinterpret_cast<const WV*>(x + m);
#pragma unroll
            for (int j = 0; j < WN; ++j) take(wide_get(a, j), true);
        } else {
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const bool have = m + j < m1;
                take(have ? x[m + j] : T(0), have);
            }
        }
    }
    __syncthreads();
    const unsigned int kept = min(min(buf_next, buf_valid), (unsigned int)SELECT_LDS_CAND);
    if (kept) {
        if (threadIdx.x == 0) g_base = atomicAdd(&cand_n[row], (unsigned long long)kept);
        __syncthreads();
        const unsigned long long gb = g_base;
        for (unsigned int i = threadIdx.x; i < kept; i += FIVEEQ_BLOCK)
            if ((int64_t)(gb + i) < cap) out[gb + i] = buf[i];
    }
}

// 6d.  PICK.  The order statistics themselves, read off the candidates WITHOUT sorting them.  ranks[row][q] is where target q
// sits among the row's candidates taken in ascending order — integer bookkeeping on the histogram, done by the host BEFORE
// the selection pass ran (for the order statistic of index i in bin b: the members of marked bins below b, plus i - cdf[b-1]) —
// so selection and pick run back to back with no host round trip between them.  One 1024-thread workgroup per (row, target)
// finds the candidate of that rank by RADIX SELECTION on the order-preserving integer image of the values: 11 bits per pass
// from the top, a 2048-bin LDS histogram of the candidates that share the target's prefix so far (wave-aggregated: in the
// top pass every candidate shares one bin), one wave then walks the bins to the one holding the rank.  3 (fp32) or 6 (fp64)
// passes over a few thousand to a few hundred thousand L2-resident values.  pool[row][seg][width] holds the candidates as
// they arrived: one segment (this rank's cand buffer), or one per rank on the root; seg_n[row][seg] valid entries each.
// Out: picked[row][n_targets] fp64; NaN when the rank is negative or not below the row's number of candidates.
constexpr int PICK_BLOCK = 1024;
template <typename T> struct SortKey;
template <> struct SortKey<float> {
    using U = unsigned int;
    static constexpr int BITS = 32;
    static __device__ __forceinline__ U of(float v) {
        const U u = __float_as_uint(v);
        return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
    }
    static __device__ __forceinline__ double back(U k) {
        return (double)__uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
    }
};
template <> struct SortKey<double> {
    using U = unsigned long long;
    static constexpr int BITS = 64;
    static __device__ __forceinline__ U of(double v) {
        const U u = (U)__double_as_longlong(v);
        return u ^ ((u >> 63) ? 0xffffffffffffffffull : 0x8000000000000000ull);
    }
    static __device__ __forceinline__ double back(U k) {
        return __longlong_as_double((long long)(k ^ ((k >> 63) ? 0x8000000000000000ull : 0xffffffffffffffffull)));
    }
};

constexpr int PICK_DIGIT = 11;                      // bits per radix pass: 2048 LDS bins; 3 passes for fp32, 6 for fp64
template <typename T>
__global__ __launch_bounds__(PICK_BLOCK) void select_pick_kernel(
    const int n_seg, const int64_t width, const T* __restrict__ pool, const unsigned long long* __restrict__ seg_n,
    const int n_targets, const long long* __restrict__ ranks /* [rows][n_targets] */, double* __restrict__ picked) {
    using K = SortKey<T>;
    using U = typename K::U;
    __shared__ unsigned int hist[1 << PICK_DIGIT];
    __shared__ long long rank_s;                 // remaining rank within the current prefix; < 0: no such candidate
    __shared__ U prefix_s;
    const int64_t row = blockIdx.x;
    const int q = blockIdx.y;                    // ONE target per workgroup: (rows x targets) workgroups share the chip
    const int Q = n_targets;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T* const x = pool + row * n_seg * width;
    if (threadIdx.x == 0) {
        long long total = 0;                                  // a segment holds at most `width` stored candidates, whatever was FOUND
        for (int g = 0; g < n_seg; ++g) total += min((long long)seg_n[row * n_seg + g], (long long)width);
        long long r = ranks[row * Q + q];
        if (r < 0 || r >= total) r = -1;
        rank_s = r;
        prefix_s = (U)0;
    }
    __syncthreads();
    if (rank_s >= 0) {                                                   // workgroup-uniform
        for (int hi = K::BITS; hi > 0;) {
            const int lo = hi > PICK_DIGIT ? hi - PICK_DIGIT : 0;
            const int nb = hi - lo;
            for (int i = threadIdx.x; i < (1 << nb); i += PICK_BLOCK) hist[i] = 0u;
            __syncthreads();
            const U prefix = prefix_s;
            for (int g = 0; g < n_seg; ++g) {
                const int64_t cnt = min((int64_t)seg_n[row * n_seg + g], width);
                const T* xs = x + g * width;
                // whole waves iterate together (wave_lds_add is a wave-level operation)
                auto tally = [&](const T v, const bool have) {
                    const U key = K::of(v);
                    const bool match = have && (hi >= K::BITS || (key >> hi) == prefix);
                    const unsigned int b = (unsigned int)((key >> lo) & (U)((1u << nb) - 1u));
                    wave_lds_add(&hist[b], 1u, match ? b : ~0u);
                };
                int64_t base = (int64_t)wave * 64;
                for (; base + 3 * PICK_BLOCK + 64 <= cnt; base += 4 * PICK_BLOCK) {      // four independent loads in flight
                    const T v0 = xs[base + lane], v1 = xs[base + PICK_BLOCK + lane], v2 = xs[base + 2 * PICK_BLOCK + lane],
                            v3 = xs[base + 3 * PICK_BLOCK + lane];
                    tally(v0, true);
                    tally(v1, true);
                    tally(v2, true);
                    tally(v3, true);
                }
                for (; base < cnt; base += PICK_BLOCK) {
                    const bool have = base + lane < cnt;
                    tally(have ? xs[base + lane] : T(0), have);
                }
            }
            __syncthreads();
            if (wave == 0) {                                             // which bin holds the rank?  lane l owns bins [l*per, (l+1)*per)
                const int per = (1 << nb) / 64;                          // 32 (11 bits) or 16 (10 bits)
                const long long r = rank_s;
                unsigned int mine = 0u;
                for (int i = 0; i < per; ++i) mine += hist[lane * per + i];
                unsigned int incl = mine;                                // inclusive prefix sum over the lanes
#pragma unroll
                for (int sh = 1; sh < 64; sh <<= 1) {
                    const unsigned int up = __shfl_up(incl, sh);
                    if (lane >= sh) incl += up;
                }
                const long long before = (long long)incl - mine;
                if (r >= before && r < (long long)incl) {                // exactly one lane: the counts sum to more than r
                    long long rem = r - before;
                    int b = lane * per;
                    while (rem >= (long long)hist[b]) rem -= hist[b++];
                    rank_s = rem;
                    prefix_s = (nb < K::BITS ? (prefix << nb) : (U)0) | (U)b;
                }
            }
            __syncthreads();
            hi = lo;
        }
    }
    if (threadIdx.x == 0) picked[row * Q + q] = rank_s < 0 ? __builtin_nan("") : K::back(prefix_s);
}

// ---------------------------------------------------------------------------------
// Diagnostic — STREAM copy with the step kernel's access shape (8 B per lane), used to
// measure achievable bandwidth and to calibrate the FETCH_SIZE / WRITE_SIZE counters.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(FIVEEQ_BLOCK) void stream_copy_kernel(const int64_t n, const double* __restrict__ src,
                                                                   double* __restrict__ dst) {
    // four independent 8-byte loads in flight per lane, like the step kernel's row loads
    const int64_t stride = (int64_t)gridDim.x * FIVEEQ_BLOCK;
    int64_t i = (int64_t)blockIdx.x * FIVEEQ_BLOCK + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const double v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
        dst[i] = v0;
        dst[i + stride] = v1;
        dst[i + 2 * stride] = v2;
        dst[i + 3 * stride] = v3;
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------
// Diagnostic — evaluate one of the hand-written math primitives over an array, so that tests can
// pin each of them against a CPU libm to the ulp, independently of the model.
// op: 0 expm1 (x <= 0), 1 exp, 2 log (x > 0), 3 sqrt (x > 0), 4 reciprocal (x > 0).
// fp32 only: op + 8 evaluates the PACKED twin (two members per lane) of the same primitive on the element pairs
// (x[2i], x[2i+1]) — it must give the scalar routine's bits (n even).
// ---------------------------------------------------------------------------------
template <typename V>
__device__ __forceinline__ V math_probe_eval(const int op, const V v) {
    switch (op) {
        case 0: return fe_expm1_neg(v);
        case 1: return fe_exp(v);
        case 2: return fe_log(v);
        case 3: return fe_sqrt(v);
        default: return fe_rcp(v);
    }
}
template <typename T>
__global__ __launch_bounds__(FIVEEQ_BLOCK) void math_probe_kernel(const int op, const int64_t n,
                                                                  const T* __restrict__ x, T* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * FIVEEQ_BLOCK + threadIdx.x;
    if constexpr (sizeof(T) == 4) {
        if (op >= 8) {
            if (2 * i + 1 < n) {
                const float2v r = math_probe_eval(op - 8, float2v{x[2 * i], x[2 * i + 1]});
                y[2 * i] = r.x;
                y[2 * i + 1] = r.y;
            }
            return;
        }
    }
    if (i >= n) return;
    y[i] = math_probe_eval(op, x[i]);
}

// Same copy with 16 B per lane (the widest access, 1 KiB per wave-instruction) and four loads in
// flight: the best plain copy this box does, quoted beside the 8 B/lane figure.  n must be even and
// both pointers 16-byte aligned (checked on the host).
__global__ __launch_bounds__(FIVEEQ_BLOCK) void stream_copy_wide_kernel(const int64_t n2, const double2* __restrict__ src,
                                                                        double2* __restrict__ dst) {
    // each workgroup copies contiguous 16 KiB tiles (4 x 256 lanes x 16 B), four loads in flight per lane
    const int64_t tile = 4 * FIVEEQ_BLOCK;
    for (int64_t base = (int64_t)blockIdx.x * tile; base < n2; base += (int64_t)gridDim.x * tile) {
        const int64_t i = base + threadIdx.x;
        if (base + tile <= n2) {
            const double2 v0 = src[i], v1 = src[i + FIVEEQ_BLOCK], v2 = src[i + 2 * FIVEEQ_BLOCK], v3 = src[i + 3 * FIVEEQ_BLOCK];
            dst[i] = v0;
            dst[i + FIVEEQ_BLOCK] = v1;
            dst[i + 2 * FIVEEQ_BLOCK] = v2;
            dst[i + 3 * FIVEEQ_BLOCK] = v3;
        } else {
            for (int64_t j = i; j < n2; j += FIVEEQ_BLOCK) dst[j] = src[j];
        }
    }
}

// The copy with the NON-TEMPORAL policy on both sides: 8 B per lane, four loads in flight, one workgroup per 8 KiB tile
// (tools/microbench/hbm_rates.hip: the fastest copy of the shapes tried on MI355X, 6.3 TB/s against 5.6-5.8 for the default
// policy at either width — nothing of a 2 GiB copy is worth keeping in the Infinity Cache).  n a multiple of 1024.
__global__ __launch_bounds__(FIVEEQ_BLOCK) void stream_copy_nt_kernel(const int64_t n, const double* __restrict__ src,
                                                                      double* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * (4 * FIVEEQ_BLOCK) + threadIdx.x;
    if (i + 3 * FIVEEQ_BLOCK >= n) return;
    const double v0 = __builtin_nontemporal_load(src + i), v1 = __builtin_nontemporal_load(src + i + FIVEEQ_BLOCK),
                 v2 = __builtin_nontemporal_load(src + i + 2 * FIVEEQ_BLOCK), v3 = __builtin_nontemporal_load(src + i + 3 * FIVEEQ_BLOCK);
    __builtin_nontemporal_store(v0, dst + i);
    __builtin_nontemporal_store(v1, dst + i + FIVEEQ_BLOCK);
    __builtin_nontemporal_store(v2, dst + i + 2 * FIVEEQ_BLOCK);
    __builtin_nontemporal_store(v3, dst + i + 3 * FIVEEQ_BLOCK);
}

// A kernel that does nothing for a known time: ONE wave, `iters` dependent fp64 FMAs (~3.5 ns each).  The host uses two of
// them to find out whether two HIP streams really run side by side (streams that share a hardware queue do not:
// fiveeqscm_amd/tuning.py, concurrent_side_streams).  Bounded by construction: the host caps iters.
__global__ __launch_bounds__(64) void busy_kernel(const int64_t iters, double* __restrict__ out) {
    double x = 1.0e-9 * (double)threadIdx.x;
    for (int64_t i = 0; i < iters; ++i) x = fe_fma(x, 0.999999, 1.0e-9);
    if (threadIdx.x == 0) out[0] = x;
}

}  // namespace fiveeq
