/*
 * fiveeq.h — C ABI of the MI355X (gfx950) ensemble engine for the five-equation
 * FaIR simple climate model.   Library: fiveeqscm_amd/csrc/libfiveeq_hip.so
 *
 * WHAT THIS REPLACES IN THE REFERENCE (stujen/fiveEqSCM @ v0)
 *   The reference has NO native code and NO FFI: its whole
 *   runtime is the Python function
 *       calculate_hfc_conc(emissions, time, lifetime)   U_FaIR/concentrations.py:4-5
 *   (duplicate: example/concentrations.py:4-5).  The five equations its README
 *   announces (README.md:2,6,8) — pool decay R_i, iIRF->alpha, concentration C,
 *   forcing F, two thermal boxes T_j — are not implemented there; the only trace
 *   of the intended function split is the list of names at .coveragerc:12-19
 *   (step_conc, step_forc, step_temp, g_1, g_0, alpha_val, k_q).  Each entry
 *   point below cites the reference item it stands behind, or says "new".
 *
 * CONVENTIONS
 *   - Plain C: pointers, sizes, one POD struct.  No torch / C++ types.
 *   - Every `dev` pointer is DEVICE memory owned by the caller (the Python host
 *     passes torch-ROCm tensor data_ptr()s).  The library allocates nothing on
 *     the device, never synchronises the stream (launches are asynchronous) and is
 *     safe to call concurrently from several threads on different streams / devices.
 *     ALL the state it keeps: the thread-local error string (fiveeq_last_error), and TWO
 *     process-wide words, the fp32 packing switch of fiveeq_set_f32_packing and the row cache
 *     policy of fiveeq_set_row_policy — atomics that every call reads once, so a call in flight
 *     while another thread flips one runs entirely with the old or entirely with the new setting
 *     (every setting gives the same bits).
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - Return value: 0 = success; <0 = error (FIVEEQ_E_*), message retrievable
 *     with fiveeq_last_error() on the same thread.  Arguments are validated on
 *     the host BEFORE any launch: a bad shape never reaches the GPU.
 *   - Struct-of-arrays over ensemble members.  A "row" is one quantity for all
 *     members: row k of array X starts at X + k*ld; member m is element m of the
 *     row (0 <= m < n_members <= ld).  `ld` (leading dimension, in elements)
 *     lets a caller run a sub-range of a larger allocation — e.g. the host's chunk-major schedule
 *     for ensembles larger than the Infinity Cache: all steps for members [0, c), then [c, 2c), ...
 *
 * MODEL STEP (identical arithmetic in every kernel; fp64 or fp32)
 *     T_old = S_0 + S_1
 *     per gas g:  G_a  = (sum_i R_gi) / c_g ;   G_u = cumE_g - G_a
 *                 iIRF = min(r0 + rC*G_u + rT*T_old + ra*G_a, iirf_max)
 *                 alpha= g0 * exp(iIRF / g1)
 *                 R_gi+= expm1(-dt/(alpha*tau_i)) * (R_gi - a_i*c_g*E_g*alpha*tau_i)
 *                 C_g  = C0_g + sum_i R_gi
 *                 F   += f1*ln(C_g/C0_g) + f2*(C_g-C0_g) + f3*(sqrt(C_g)-sqrt(C0_g))
 *     F += F_ext ;  S_j += expm1(-dt/d_j) * (S_j - q_j*F) ;  T = S_0 + S_1
 */
#ifndef FIVEEQ_H
#define FIVEEQ_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FIVEEQ_ABI_VERSION   11
#define FIVEEQ_MAX_GAS       3
#define FIVEEQ_MAX_POOLS     4
#define FIVEEQ_N_BOX         2
#define FIVEEQ_DRIVE_STRIDE  8   /* elements per step in the drive table, below */
#define FIVEEQ_LHS_MAX_TOTAL (1LL << 28)   /* largest Latin-hypercube design (members): stratum + jitter stays an exact fp64 sum */

#define FIVEEQ_OK             0
#define FIVEEQ_E_INVALID     -1  /* bad argument (NULL pointer, size, range) */
#define FIVEEQ_E_UNSUPPORTED -2  /* pool layout not instantiated */
#define FIVEEQ_E_HIP         -3  /* a HIP runtime call failed */

/* Shared (not per-member) parameters of one gas.  Active pools are the first
 * n_pools entries of a[] / tau[].  g0 and g1 are the alpha-closure constants
 * (host helpers g_0 / g_1; names from .coveragerc:15-16 of the reference). */
typedef struct fiveeq_gas {
    double  a[FIVEEQ_MAX_POOLS];    /* pool fractions                       */
    double  tau[FIVEEQ_MAX_POOLS];  /* pool time-scales, years              */
    double  g0, g1;                 /* alpha = g0 * exp(iIRF / g1)          */
    double  ra;                     /* iIRF sensitivity to own burden G_a   */
    double  C0;                     /* pre-industrial concentration         */
    double  emis2conc;              /* c_g: concentration units per emission unit */
    double  f[3];                   /* forcing coefficients: log, linear, sqrt */
    int32_t n_pools;                /* 1..4                                  */
    int32_t reserved;
} fiveeq_gas;

/* Whole shared model.  Always given in double; the f32 entry points round it.
 * ACCURACY OF THE _f32 ENTRY POINTS, against 50-digit arithmetic over the 24 golden members x 750 steps
 * (tests/test_golden_fiveeq.py): C within 2.9e-6 relative, T within 1.7e-5 (test bounds 5e-6 / 3e-5 + 2e-6 K).  What bounds
 * it is the fp32 rounding of the STATE carried through 750 steps (~13 ulp of C at the end), not the transcendental forms:
 * with a two-step expm1 reduction, a Newton step on the reciprocal and an fdlibm-style logarithm in place of the hardware
 * v_rcp_f32 / v_log_f32 forms the same members changed 48 of 840 stored values and no worst case, for +21 % on the
 * time-fused kernel (profiles/r05/fp32_math_ab.txt; the variant was measured at commit 51b82a6 and not kept).  A run that
 * needs more digits than that runs the _f64 entry points. */
typedef struct fiveeq_model {
    fiveeq_gas gas[FIVEEQ_MAX_GAS];
    double  d[FIVEEQ_N_BOX];        /* thermal-box time-scales, years       */
    double  iirf_max;               /* clip on iIRF (e.g. 97)               */
    double  dt;                     /* step length, years                   */
    int32_t n_gas;                  /* 1..3                                  */
    int32_t reserved;
} fiveeq_model;

/* Array shapes used below (G = n_gas, SP = sum of n_pools over gases):
 *   drive   dev [n_steps][8]   shared by all members, per step:
 *                              [0..2] E_g (emission rate), [3..5] cumulative emissions BEFORE
 *                              the step, [6] F_ext, [7] OUTPUT ROW of this step: the step's C and
 *                              T are stored at row k = (int)drive[t][7] of C_traj / T_traj if
 *                              0 <= k < n_rows, and not stored otherwise (store every step:
 *                              drive[t][7] = t, n_rows = n_steps)
 *   r       dev [3*G][ld]      rows g*3+0/1/2 = r0, rC, rT of gas g (per member)
 *   q       dev [2][ld]        thermal-box coefficients (per member)
 *   R       dev [SP][ld]       pool contents, gas-major, in/out
 *   S       dev [2][ld]        thermal-box temperatures, in/out
 *   C_traj  dev [n_rows][G][ld]  concentrations of the stored steps (may be NULL)
 *   T_traj  dev [n_rows][ld]     temperature of the stored steps    (may be NULL)
 *   T_stats dev [W][n_steps][4]  fp64 (also for the f32 entry points), W = fiveeq_stats_waves(n_members):
 *                              per wave of 64 members and per step (sum T, sum T^2, min T, max T);
 *                              folding over W gives the ensemble moments of every step without a
 *                              stored trajectory (may be NULL).  Every step in the range writes its
 *                              record in all W wave rows.  Wave-major on purpose: a call on the member
 *                              sub-range starting at member m0 (a multiple of 64) addresses its part of
 *                              a larger buffer as T_stats + (m0/64)*n_steps*4.
 */

/* new — library identification */
int         fiveeq_abi_version(void);
const char *fiveeq_last_error(void);
/* new — sha256 (hex) of the three sources this library was compiled from, concatenated in this order:
 * fiveeq_capi.hip, fiveeq_device.hpp, include/fiveeq.h — stamped by csrc/Makefile ("unstamped" otherwise).  A binding that
 * sits next to those sources recomputes it and refuses a library built from other text (fiveeqscm_amd/_capi.py). */
const char *fiveeq_source_hash(void);
/* new — the experiment knobs this library was compiled with (-DFIVEEQ_STEP_WAVES=..., -DFIVEEQ_FUSED_CHUNK=..., non-default
 * block sizes ...), space-separated; "" for the product build. */
const char *fiveeq_build_flags(void);
/* new — sizeof(fiveeq_model) as the library was compiled, for binding self-checks */
int64_t     fiveeq_sizeof_model(void);
/* new — 1 if (n_gas, n_pools[]) has a compiled kernel, else 0 */
int         fiveeq_layout_supported(int32_t n_gas, const int32_t *n_pools);
/* new — W of T_stats: ceil(n_members / 64) */
int64_t     fiveeq_stats_waves(int64_t n_members);

/* ONE TIMESTEP, ONE LAUNCH: the north-star hot path.  Stands where the
 * reference intended step_conc + alpha_val + step_forc + step_temp
 * (.coveragerc:12-14,17 — names only; no reference code exists).
 * Reads state/params from HBM, writes state back, writes the step's C, T rows and stats. */
int fiveeq_step_f64(const fiveeq_model *model, int64_t n_members, int64_t ld,
                    const double *drive, int32_t n_steps, int32_t t,
                    const double *r, const double *q, double *R, double *S,
                    double *C_traj, double *T_traj, int32_t n_rows, double *T_stats, void *stream);
int fiveeq_step_f32(const fiveeq_model *model, int64_t n_members, int64_t ld,
                    const float *drive, int32_t n_steps, int32_t t,
                    const float *r, const float *q, float *R, float *S,
                    float *C_traj, float *T_traj, int32_t n_rows, double *T_stats, void *stream);

/* Steps t_begin <= t < t_end as (t_end - t_begin) launches of the kernel above,
 * enqueued back-to-back on `stream` from C (no Python per step). */
int fiveeq_run_f64(const fiveeq_model *model, int64_t n_members, int64_t ld,
                   const double *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                   const double *r, const double *q, double *R, double *S,
                   double *C_traj, double *T_traj, int32_t n_rows, double *T_stats, void *stream);
int fiveeq_run_f32(const fiveeq_model *model, int64_t n_members, int64_t ld,
                   const float *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                   const float *r, const float *q, float *R, float *S,
                   float *C_traj, float *T_traj, int32_t n_rows, double *T_stats, void *stream);

/* The same launch sequence captured once into a hipGraph ("plan") and replayed:
 * removes per-launch host cost for small ensembles.  The plan bakes in the
 * pointers; they must stay valid until fiveeq_plan_destroy. */
int fiveeq_plan_create_f64(const fiveeq_model *model, int64_t n_members, int64_t ld,
                           const double *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                           const double *r, const double *q, double *R, double *S,
                           double *C_traj, double *T_traj, int32_t n_rows, double *T_stats, void **plan_out);
int fiveeq_plan_create_f32(const fiveeq_model *model, int64_t n_members, int64_t ld,
                           const float *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                           const float *r, const float *q, float *R, float *S,
                           float *C_traj, float *T_traj, int32_t n_rows, double *T_stats, void **plan_out);
int fiveeq_plan_launch(void *plan, void *stream);
int fiveeq_plan_destroy(void *plan);

/* TIME-FUSED variant: one launch advances t_begin..t_end with
 * the member's state held in registers; only the stored C/T rows and the stats are written per
 * step.  Same arithmetic, bit-identical results to the per-step path. */
int fiveeq_run_fused_f64(const fiveeq_model *model, int64_t n_members, int64_t ld,
                         const double *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                         const double *r, const double *q, double *R, double *S,
                         double *C_traj, double *T_traj, int32_t n_rows, double *T_stats, void *stream);
int fiveeq_run_fused_f32(const fiveeq_model *model, int64_t n_members, int64_t ld,
                         const float *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                         const float *r, const float *q, float *R, float *S,
                         float *C_traj, float *T_traj, int32_t n_rows, double *T_stats, void *stream);

/* K STEPS PER LAUNCH: the time-fused kernel over consecutive spans of k_steps
 * — state and parameters cross HBM once per k_steps instead of every step, A_K = w(G+1)[stored rows]
 * + w(2SP+3G+6)/k_steps.  The form for ensembles too small to hide the ~2 us dependent-launch boundary
 * behind one step of HBM traffic (10k members: 2.6 us per launch against 0.2 us of traffic).
 * Bit-identical results to the per-step path. */
int fiveeq_run_ksteps_f64(const fiveeq_model *model, int64_t n_members, int64_t ld,
                          const double *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                          const double *r, const double *q, double *R, double *S,
                          double *C_traj, double *T_traj, int32_t n_rows, double *T_stats,
                          int32_t k_steps, void *stream);
int fiveeq_run_ksteps_f32(const fiveeq_model *model, int64_t n_members, int64_t ld,
                          const float *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                          const float *r, const float *q, float *R, float *S,
                          float *C_traj, float *T_traj, int32_t n_rows, double *T_stats,
                          int32_t k_steps, void *stream);

/* SMALL ENSEMBLES (BASELINE configs[1], 10k CO2-only members): the time-fused step with ONE MEMBER
 * SPREAD OVER SEVERAL LANES.  An ensemble of fewer waves than the chip has SIMDs (1024) is bound by the number of
 * instructions one wave issues per step; with lanes_per_member = 4 (pool layouts {4}: a lone 4-pool gas) lane 4m + i carries
 * pool i of member m — one expm1 chain per wave-step instead of four, the two sums over pools folded with DPP moves in the
 * per-step kernel's order ((R0 + R1) + R2) + R3 — and the shared model stays in registers instead of LDS.  4x the waves of
 * fiveeq_run_fused_*, 16 members per wave.  lanes_per_member: 4, 8 (new, ABI v11: the 4 + 1 + 1 layout on an OCTET of lanes —
 * gas 0's four pools on lanes 0-3, the single pools of gases 1 and 2 on lanes 4 and 5: every lane runs one gas's closure, one
 * expm1 chain and one forcing, a third of the one-lane form's instructions per wave; T_stats must be NULL for it), 1 (one member
 * per lane with the model in registers: every compiled layout, several gases included) or 0 = the widest form the layout has
 * (fiveeq_small_lanes; the one-lane form when a 4 + 1 + 1 run asks for statistics).  One launch for the
 * whole span; C_traj, T_traj, the row map and T_stats as in the other entry points (the statistics records are the fused
 * kernel's bit for bit).  Same arithmetic operation for operation: bit-identical results to the per-step path.
 * Ahead of the fused kernel while the ensemble is launch-bound — about 64 members per CU for the quad form, a few hundred
 * thousand members for the one-lane form (10k members, us per step: CO2-only fp64 0.42 against 0.73, three gases fp64 0.94
 * against 1.38, fp32 0.51 against 1.16). */
int fiveeq_run_small_f64(const fiveeq_model *model, int64_t n_members, int64_t ld,
                         const double *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                         const double *r, const double *q, double *R, double *S,
                         double *C_traj, double *T_traj, int32_t n_rows, double *T_stats,
                         int32_t lanes_per_member, void *stream);
int fiveeq_run_small_f32(const fiveeq_model *model, int64_t n_members, int64_t ld,
                         const float *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                         const float *r, const float *q, float *R, float *S,
                         float *C_traj, float *T_traj, int32_t n_rows, double *T_stats,
                         int32_t lanes_per_member, void *stream);
/* new (ABI v11) — THE COMPENSATED fp32 FORM of the time-fused kernel (opt-in; BASELINE configs[4]'s natural mode: 100M fp32
 * members with the state in registers).  Two changes against fiveeq_run_fused_f32 / fiveeq_run_ksteps_f32 / _fused_bins_f32,
 * whose arguments it takes (k_steps: steps per launch, >= t_end - t_begin = one launch; bin_ring = NULL: no histogram ring and
 * lo / hi / n_bins / ring_rows are ignored):
 *   (1) every POOL carries a second register word holding the rounding error of its own update, fed back into the next one
 *       (Kahan's summation with the increment's product fused into the first add): +3 instructions per pool and step, NO HBM
 *       bytes — the words start at zero in every launch and are dropped at its end, so R / S in memory stay plain fp32 rows (a
 *       run relaunched every k_steps steps loses at most one rounding per pool and launch).  The two thermal boxes are not
 *       compensated: once (2) is in place their rounding is 4e-7 of T;
 *   (2) the forcing is computed from the EXCESS C - C0 = sum_i R_i, not from the rounded C: ln(C/C0) as log1p((C - C0)/C0),
 *       sqrt(C) - sqrt(C0) as (C - C0) / (sqrt(C) + sqrt(C0)).  In fp32 the default form loses the small excess of the first
 *       decades to the rounding of C itself, and THAT is what bounds T in fp32, not the state.
 * Worst relative error against 50-digit arithmetic over the 24 golden members x 750 steps: C 2.9e-6 -> 1.8e-7 (the rounding of
 * the stored C itself), T (1e-2 K floor) 1.7e-5 -> 7e-7; cost on the VALU-bound fused kernel: profiles/r06/fp32_compensated.txt.
 * It is its own arithmetic: results are NOT bit-identical to the default forms (which stay bit-identical among themselves), and
 * the kernels that keep the state in HBM (step / run / run_bins / plan) do not have it. */
int fiveeq_run_fused_comp_f32(const fiveeq_model *model, int64_t n_members, int64_t ld,
                              const float *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                              const float *r, const float *q, float *R, float *S,
                              float *C_traj, float *T_traj, int32_t n_rows, double *T_stats, int32_t k_steps,
                              double lo, double hi, int32_t n_bins, uint16_t *bin_ring, int32_t ring_rows, void *stream);
/* new (ABI v11) — the same compensated arithmetic on the SMALL-ENSEMBLE kernel (one member per lane, the model in registers; every
 * compiled layout): what a launch-bound fp32 ensemble takes instead of the fused kernel (10k three-gas members: about half the
 * time per step).  One launch for the whole span; arguments of fiveeq_run_fused_f32; the words are dropped at the end of the call. */
int fiveeq_run_small_comp_f32(const fiveeq_model *model, int64_t n_members, int64_t ld,
                              const float *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                              const float *r, const float *q, float *R, float *S,
                              float *C_traj, float *T_traj, int32_t n_rows, double *T_stats, void *stream);
/* lanes per member of the widest small-ensemble form compiled for (n_gas, n_pools[]): 4 (a lone 4-pool gas), 8 (4 + 1 + 1),
 * 1 (every other compiled layout), or 0 = the layout has no kernel at all */
int32_t fiveeq_small_lanes(int32_t n_gas, const int32_t *n_pools);

/* The fp32 entry points (step / run / run_fused / run_ksteps / run_*bins / plan_create _f32) compute TWO members per
 * lane with packed fp32 instructions and 8-byte row accesses whenever the rows allow it (ld even, every row pointer
 * 8-byte aligned, n_members >= 2), and one member per lane otherwise; both give the same bits.  This switch forces the
 * one-member-per-lane kernels (on = 0) or re-enables packing (on != 0, the default); returns the previous setting.
 * Process-wide (an atomic word, see CONVENTIONS); meant for measurements and tests. */
int fiveeq_set_f32_packing(int on);

/* CACHE POLICY OF THE PER-STEP KERNEL'S ROWS (new).  The step / run / plan_create entry points read every state and parameter
 * row once per step.  While those rows fit the 256 MiB Infinity Cache they are served from it at the next step (the default
 * policy; what a chunk-major schedule arranges for large ensembles).  A launch whose rows cannot survive until the next step
 * takes the STREAMED form instead: the same kernel with non-temporal loads and stores (bit-identical results; -4.5 % per step
 * at 8M fp64 members, -9 % at 4M; +11...13 % — the wrong form — on a cache-resident 1-2M-member ensemble).
 * FIVEEQ_ROWS_AUTO (the default) decides per call: streamed when n_members x (SP + 2 + 3G + 2) words >= the cache AND
 * ld x the same >= twice the cache (ld = the row length: the whole ensemble this call's members are a sub-range of).
 * fiveeq_set_row_policy forces one form process-wide (an atomic word read once per call, see CONVENTIONS; returns the previous
 * setting, or FIVEEQ_E_INVALID for a value that is none of the three) — meant for measurements and tests.
 * fiveeq_rows_streamed: 1 / 0 = the form a per-step launch of that shape would take now.  The stored C / T rows are written
 * non-temporally in every form (written once, never re-read by a stepping kernel).  The streamed-histogram per-step form
 * (fiveeq_run_bins_*) always takes the default policy. */
#define FIVEEQ_ROWS_CACHED      0
#define FIVEEQ_ROWS_STREAMED    1
#define FIVEEQ_ROWS_AUTO        2
int fiveeq_set_row_policy(int32_t policy);
int fiveeq_rows_streamed(int32_t n_gas, const int32_t *n_pools, int64_t n_members, int64_t ld, int32_t word_bytes);

/* CONCENTRATION-DRIVEN (inverse) mode (the reference's module name
 * `concentrations` hints at it, no reference code exists).  drive[t][0..2] hold the TARGET
 * concentration of each gas at the end of step t (shared by all members; columns 3..5 unused);
 * cumE dev [G][ld] is per-member cumulative-emission state (in/out, start at 0); the emission rate
 * that reaches the target is diagnosed per member and step from the same pool equations and
 * written to E_traj dev [n_rows][G][ld] (row map as above); pools, boxes, T and T_stats advance
 * exactly as in the forward path.  One launch for the whole span (time-fused form). */
int fiveeq_run_inverse_f64(const fiveeq_model *model, int64_t n_members, int64_t ld,
                           const double *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                           const double *r, const double *q, double *R, double *S, double *cumE,
                           double *E_traj, double *T_traj, int32_t n_rows, double *T_stats, void *stream);
int fiveeq_run_inverse_f32(const fiveeq_model *model, int64_t n_members, int64_t ld,
                           const float *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                           const float *r, const float *q, float *R, float *S, float *cumE,
                           float *E_traj, float *T_traj, int32_t n_rows, double *T_stats, void *stream);

/* Ensemble form of the reference's one function,
 *   calculate_hfc_conc(emissions, time, lifetime) = emissions[0]*exp(-time)
 * (U_FaIR/concentrations.py:4-5): out[k][m] = e0[m] * exp(-time[k]).
 *   e0 dev [n_members], time dev [n_time], out dev [n_time][ld]. */
int fiveeq_hfc_conc_f64(int64_t n_members, int64_t ld, int32_t n_time,
                        const double *e0, const double *time, double *out, void *stream);

/* new — fixed-bin histograms of stored rows, for all-timestep percentiles with a tiny exchange
 * (SURVEY.md section 8e-ii): hist[row][b] += number of members with lo + b*w <= rows[row][m] < lo + (b+1)*w,
 * w = (hi - lo)/n_bins; values outside [lo, hi) are counted in the edge bins, NaNs are skipped.
 * THE BIN RULE, shared bit for bit by every entry point that bins a value (these, the bin indices of fiveeq_run_*bins_*,
 * the summary's selection): with inv_w = n_bins/(hi - lo),
 *   _f64:  bin = trunc(clamp((x - lo) * inv_w, 0, n_bins - 1))                        evaluated in fp64
 *   _f32:  bin = trunc(clamp(fma(x, (float)inv_w, (float)(-lo*inv_w)), 0, n_bins - 1)) evaluated in fp32 (one FMA per
 *          member).  Monotone in x like the fp64 formula; against it a member changes bin only within
 *          ~2^-23 * max(|lo|, |hi|) * inv_w of a bin edge (in bins): 2^-12 bin for a range that starts near zero such as
 *          temperature anomalies, 0.01 bin for lo = 280, hi = 295, n_bins = 4096 — express rows in absolute units far from zero
 *          as anomalies, or keep them in fp64, if that matters.
 *   rows dev [n_rows][ld] (e.g. T_traj), hist dev [n_rows][n_bins] uint64, ACCUMULATED INTO (zero it first;
 *   several shards / calls may add into the same histogram), 1 <= n_bins <= 4096, n_rows <= 65535. */
int fiveeq_hist_rows_f64(int32_t n_rows, int64_t n_members, int64_t ld, const double *rows,
                         double lo, double hi, int32_t n_bins, uint64_t *hist, void *stream);
int fiveeq_hist_rows_f32(int32_t n_rows, int64_t n_members, int64_t ld, const float *rows,
                         double lo, double hi, int32_t n_bins, uint64_t *hist, void *stream);
/* new — the END-OF-RUN SUMMARY as HIP passes (form (i): exact percentiles of T at selected output
 * times over ALL members by SELECTION, so that the ensemble does not travel; host side: fiveeqscm_amd/distributed.py).
 * All three read rows dev [n_rows][ld] once, 16 bytes per lane and load.
 *
 * (1) moments dev [n_rows][4] fp64 = (sum, sum of squares, min, max) of each row's n_members values — fp64 sums of the
 *     exactly converted elements in a fixed order (same bits every run); min / max ignore NaNs, the sums propagate them.
 *     partial dev [n_rows][K][4] fp64 is workspace, K = fiveeq_row_moments_chunks(n_rows, n_members). */
int64_t fiveeq_row_moments_chunks(int32_t n_rows, int64_t n_members);
int fiveeq_row_moments_f64(int32_t n_rows, int64_t n_members, int64_t ld, const double *rows,
                           double *partial, double *moments, void *stream);
int fiveeq_row_moments_f32(int32_t n_rows, int64_t n_members, int64_t ld, const float *rows,
                           double *partial, double *moments, void *stream);
/* (2) fiveeq_hist_rows_* with a range PER ROW read from device memory — ranges dev [n_rows][2] fp64 = (lo, hi), e.g. each
 *     row's global extrema, so that no host round trip sits between the moments and the histogram; a row with hi <= lo is
 *     constant and is counted in bin 0. */
int fiveeq_hist_rows_ranged_f64(int32_t n_rows, int64_t n_members, int64_t ld, const double *rows,
                                const double *ranges, int32_t n_bins, uint64_t *hist, void *stream);
int fiveeq_hist_rows_ranged_f32(int32_t n_rows, int64_t n_members, int64_t ld, const float *rows,
                                const double *ranges, int32_t n_bins, uint64_t *hist, void *stream);
/* (3) selection.  The histogram of (2) holds exact counts and the bin rule is monotone in the value, so the order statistic
 *     of global index i lies in the bin b with cdf[b-1] <= i < cdf[b] and is the (i - cdf[b-1])-th smallest member of it.
 *     binmask dev [n_rows][ceil(n_bins/32)] uint32 marks, per row, the bins that hold wanted order statistics (bit b%32 of
 *     word b/32); the pass recomputes every member's bin with the rule and the ranges of (2), bit for bit, and appends the
 *     members of marked bins — the candidates — to cand dev [n_rows][cap] (any order within a row).  cand_n dev [n_rows]
 *     uint64, ACCUMULATED INTO, counts the candidates FOUND; those beyond cap are counted but not stored. */
int fiveeq_select_bins_f64(int32_t n_rows, int64_t n_members, int64_t ld, const double *rows,
                           const double *ranges, int32_t n_bins, const uint32_t *binmask,
                           double *cand, int64_t cap, uint64_t *cand_n, void *stream);
int fiveeq_select_bins_f32(int32_t n_rows, int64_t n_members, int64_t ld, const float *rows,
                           const double *ranges, int32_t n_bins, const uint32_t *binmask,
                           float *cand, int64_t cap, uint64_t *cand_n, void *stream);
/* (4) pick: the order statistics, read off the candidates without sorting them.  ranks dev [n_rows][n_targets] int64: where
 *     each target sits among the row's candidates in ascending order (host bookkeeping on the histogram: the members of
 *     marked bins below the target's bin, plus i - cdf[b-1]); one workgroup per (row, target) finds the candidate of that
 *     rank by radix selection.  pool dev [n_rows][n_seg][width]: the candidates as they arrived, seg_n dev [n_rows][n_seg]
 *     valid entries per segment (one segment: cand / cand_n of (3); on the root of a multi-rank exchange: one segment per
 *     rank).  A segment holds at most `width` STORED candidates: seg_n entries beyond width (cand_n of (3) counts what it
 *     found, also past cap) are taken as width.  picked dev [n_rows][n_targets] fp64; NaN where the rank is negative or not
 *     below the number of stored candidates. */
int fiveeq_select_pick_f64(int32_t n_rows, int32_t n_seg, int64_t width, const double *pool, const uint64_t *seg_n,
                           int32_t n_targets, const int64_t *ranks, double *picked, void *stream);
int fiveeq_select_pick_f32(int32_t n_rows, int32_t n_seg, int64_t width, const float *pool, const uint64_t *seg_n,
                           int32_t n_targets, const int64_t *ranks, double *picked, void *stream);

/* STREAMED HISTOGRAMS through a ring of BIN INDICES (round 3).  fiveeq_run_fused_bins_* is
 * fiveeq_run_fused_* (same arguments, same results, C_traj / T_traj / T_stats as there) that ALSO writes, for every step t of
 * the span and every member m, the histogram bin of T(t, m) — the rule of fiveeq_hist_rows_* with (hist_lo, hist_hi, n_bins),
 * bit for bit; 0xFFFF for a NaN — as one uint16 into bin_ring dev [ring_rows][ld] at row t mod ring_rows: 2 bytes per
 * member-step where a ring of T rows takes 4 or 8.  fiveeq_hist_bins then counts rows of such indices into
 * hist dev [n_rows][n_bins] uint64 (ACCUMULATED INTO).  The caller runs spans of at most ring_rows steps and drains the
 * ring between them (EnsembleEngine does, on a second stream).  The pass does not see T: per-step moments, if wanted,
 * come from T_stats. */
int fiveeq_run_fused_bins_f64(const fiveeq_model *model, int64_t n_members, int64_t ld,
                              const double *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                              const double *r, const double *q, double *R, double *S,
                              double *C_traj, double *T_traj, int32_t n_rows, double *T_stats,
                              double hist_lo, double hist_hi, int32_t n_bins,
                              uint16_t *bin_ring, int32_t ring_rows, void *stream);
int fiveeq_run_fused_bins_f32(const fiveeq_model *model, int64_t n_members, int64_t ld,
                              const float *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                              const float *r, const float *q, float *R, float *S,
                              float *C_traj, float *T_traj, int32_t n_rows, double *T_stats,
                              double hist_lo, double hist_hi, int32_t n_bins,
                              uint16_t *bin_ring, int32_t ring_rows, void *stream);
int fiveeq_hist_bins(int32_t n_rows, int64_t n_members, int64_t ld, const uint16_t *bins, int32_t n_bins,
                     uint64_t *hist, void *stream);
/* the same for the PER-STEP form: fiveeq_run_* (one launch per timestep) whose kernel also writes the bin index of T —
 * 2 bytes per member-step on top of the step's w(2 SP + 4 G + 7), where a ring of T rows adds w written + w re-read */
int fiveeq_run_bins_f64(const fiveeq_model *model, int64_t n_members, int64_t ld,
                        const double *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                        const double *r, const double *q, double *R, double *S,
                        double *C_traj, double *T_traj, int32_t n_rows, double *T_stats,
                        double hist_lo, double hist_hi, int32_t n_bins,
                        uint16_t *bin_ring, int32_t ring_rows, void *stream);
int fiveeq_run_bins_f32(const fiveeq_model *model, int64_t n_members, int64_t ld,
                        const float *drive, int32_t n_steps, int32_t t_begin, int32_t t_end,
                        const float *r, const float *q, float *R, float *S,
                        float *C_traj, float *T_traj, int32_t n_rows, double *T_stats,
                        double hist_lo, double hist_hi, int32_t n_bins,
                        uint16_t *bin_ring, int32_t ring_rows, void *stream);

/* new — shard-computable Latin hypercube: out[k][i] = u_{dim0+k}(m0 + i),
 * 0 <= i < n_members, 0 <= k < n_dim, for a design over n_total members:
 *     u_d(m) = (pi_d(m) + jitter_d(m)) / n_total
 * with pi_d a keyed bijection of [0, n_total) (cycle-walked 4-round Feistel network) and jitter a
 * 24-bit counter-based hash in (0,1): exactly one member per stratum and dimension, and a pure
 * function of (seed, d, m, n_total) — every rank computes only its own members, on its own device,
 * and gets the same design whatever the world size.  out dev [n_dim][ld] fp64.
 * 1 <= n_total <= FIVEEQ_LHS_MAX_TOTAL (2^28): up to there pi + jitter (28 + 25 bits) is an exact fp64 sum, so u lies
 * strictly inside its stratum; larger designs are refused rather than rounded onto a stratum edge.
 * Host twin (bit-identical): fiveeqscm_amd.params.lhs_rows. */
int fiveeq_lhs_rows_f64(uint64_t seed, int64_t n_total, int64_t m0, int64_t n_members,
                        int32_t dim0, int32_t n_dim, int64_t ld, double *out, void *stream);
/* the same rows into HOST memory, computed on the calling thread by the same functions (no GPU needed) */
int fiveeq_lhs_rows_host_f64(uint64_t seed, int64_t n_total, int64_t m0, int64_t n_members,
                             int32_t dim0, int32_t n_dim, int64_t ld, double *out);

/* new — diagnostic: STREAM-style copy dst[i] = src[i], i < n, with the SAME access shape as the
 * step kernel (one 8-byte element per lane, 512 B per wave-instruction, grid sized the same way).
 * Used to measure the achievable copy bandwidth on the box and to calibrate the rocprofv3
 * FETCH_SIZE / WRITE_SIZE counters on a known byte count (MI355X_MICROARCH.md, HBM section). */
int fiveeq_stream_copy_f64(int64_t n, const double *src, double *dst, void *stream);
/* the same copy with 16 B per lane (n even, pointers 16-byte aligned): the box's best plain copy */
int fiveeq_stream_copy_wide_f64(int64_t n, const double *src, double *dst, void *stream);
/* the same copy, 8 B per lane, NON-TEMPORAL loads and stores, one workgroup per 8 KiB (n a multiple of 1024): the fastest
 * plain copy measured on MI355X (tools/microbench/hbm_rates.hip) — the ceiling bench.py's hbm_resident figure is held against */
int fiveeq_stream_copy_nt_f64(int64_t n, const double *src, double *dst, void *stream);
/* new — diagnostic: ONE wave that runs `iterations` (0..1e8) dependent fp64 FMAs (~3.5 ns each) and writes one double to
 * `out`: a launch of known duration that occupies one SIMD.  Two of them on two streams take the time of one when the streams
 * run side by side and of two when they share a hardware queue — how the Python host picks the side streams of its two-stream
 * schedules (a stream that shares the caller's hardware queue costs the per-step form +12 %). */
int fiveeq_busy(int64_t iterations, double *out, void *stream);

/* new — diagnostic: y[i] = f(x[i]) with one of the kernels' own fp64 math primitives, so tests can
 * pin each against a CPU libm to the ulp.  op: 0 expm1 (x <= 0), 1 exp, 2 log (x > 0, finite normal),
 * 3 sqrt (x > 0, finite normal), 4 reciprocal (x > 0, finite normal).  _f32 only: op + 8 evaluates the PACKED twin of
 * the primitive (two members per lane, fiveeq_set_f32_packing above) on the element pairs (x[2i], x[2i+1]), n even:
 * it must return the scalar routine's bits. */
int fiveeq_math_probe_f64(int32_t op, int64_t n, const double *x, double *y, void *stream);
int fiveeq_math_probe_f32(int32_t op, int64_t n, const float *x, float *y, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FIVEEQ_H */
