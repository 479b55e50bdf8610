/*
 * fiveeq_oracle.c — plain-C fp64 restatement of the five-equation step, member loop.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load the library built from this file, as the
 * checker / the timed CPU baseline.  The product (fiveeqscm_amd/) never links it.
 *
 * PARITY STATUS: "parity unpinned" for the five equations — the reference
 * (stujen/fiveEqSCM @ v0) has no implementation, test or golden vector for them
 * (its README.md:8,10).  This file follows oracle/fiveeq_oracle.py line for line (same
 * formulas, libm exp/expm1/log/sqrt) and is cross-checked against it in
 * tests/test_oracle.py; the NumPy oracle in turn carries the analytic known-answer
 * tests.  The one reference function, emissions[0]*exp(-time)
 * (U_FaIR/concentrations.py:4-5), is restated as oracle_hfc_conc() and pinned by
 * tests/golden/hfc_conc_golden.json.
 *
 * Layout = the engine's (include/fiveeq.h): rows [k][ld] over members; drive [n_steps][8].
 * Build: oracle/Makefile  (gcc -O2 -fopenmp -shared; NO -ffast-math: libm calls stay exact).
 */
#include <math.h>
#include <stdint.h>

#define MAX_GAS 3
#define MAX_POOLS 4
#define DRIVE_STRIDE 8

typedef struct {
    double a[MAX_POOLS], tau[MAX_POOLS];
    double g0, g1, ra, C0, emis2conc, f[3];
    int32_t n_pools, reserved;
} oracle_gas;

typedef struct {
    oracle_gas gas[MAX_GAS];
    double d[2], iirf_max, dt;
    int32_t n_gas, reserved;
} oracle_model;   /* same bytes as fiveeq_model, so tests can pass one struct to both */

/* alpha_val — oracle/fiveeq_oracle.py: alpha_val() */
static double alpha_val(double G_u, double G_a, double T, double r0, double rC, double rT, double ra,
                        double g0, double g1, double iirf_max) {
    double iirf = r0 + rC * G_u + rT * T + ra * G_a;
    if (iirf > iirf_max) iirf = iirf_max;
    return g0 * exp(iirf / g1);
}

/* step_forc — oracle/fiveeq_oracle.py: step_forc() */
static double step_forc(double C, double C0, const double *f) {
    const int pos = C > 0.0;
    const double logt = pos ? log(C / C0) : 0.0;
    const double sqrtt = (pos ? sqrt(C) : 0.0) - sqrt(C0);
    return f[0] * logt + f[1] * (C - C0) + f[2] * sqrtt;
}

/* Steps t_begin..t_end-1 for members m0..m1-1.  C_traj [n_steps][G][ld] / T_traj [n_steps][ld] may be NULL. */
int oracle_run(const oracle_model *mdl, int64_t n, int64_t ld, const double *drive, int32_t n_steps,
               int32_t t_begin, int32_t t_end, const double *r, const double *q, double *R, double *S,
               double *C_traj, double *T_traj, int32_t n_threads) {
    if (!mdl || !drive || !r || !q || !R || !S || n < 1 || ld < n || t_begin < 0 || t_end > n_steps) return -1;
    const int G = mdl->n_gas;
    int off[MAX_GAS + 1];
    off[0] = 0;
    for (int g = 0; g < G; ++g) off[g + 1] = off[g] + mdl->gas[g].n_pools;
    const double em1_d0 = expm1(-mdl->dt / mdl->d[0]), em1_d1 = expm1(-mdl->dt / mdl->d[1]);
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int64_t m = 0; m < n; ++m) {
        double Rm[MAX_GAS * MAX_POOLS];
        for (int k = 0; k < off[G]; ++k) Rm[k] = R[k * ld + m];
        double S0 = S[m], S1 = S[ld + m];
        const double q0 = q[m], q1 = q[ld + m];
        for (int t = t_begin; t < t_end; ++t) {
            const double *drv = drive + (int64_t)t * DRIVE_STRIDE;
            const double T_old = S0 + S1;
            double F = 0.0;
            for (int g = 0; g < G; ++g) {
                const oracle_gas *gs = &mdl->gas[g];
                const int P = gs->n_pools;
                double *Rg = Rm + off[g];
                double sumR = 0.0;
                for (int i = 0; i < P; ++i) sumR += Rg[i];
                const double G_a = sumR / gs->emis2conc;
                const double G_u = drv[3 + g] - G_a;
                const double alpha = alpha_val(G_u, G_a, T_old, r[(3 * g) * ld + m], r[(3 * g + 1) * ld + m],
                                               r[(3 * g + 2) * ld + m], gs->ra, gs->g0, gs->g1, mdl->iirf_max);
                /* step_conc — oracle/fiveeq_oracle.py: step_conc() */
                const double E_c = drv[g] * gs->emis2conc;
                double sumN = 0.0;
                for (int i = 0; i < P; ++i) {
                    const double at = alpha * gs->tau[i];
                    const double em1 = expm1(-mdl->dt / at);
                    Rg[i] = Rg[i] + em1 * (Rg[i] - (gs->a[i] * E_c) * at);
                    sumN += Rg[i];
                }
                const double C = gs->C0 + sumN;
                F = F + step_forc(C, gs->C0, gs->f);
                if (C_traj) C_traj[((int64_t)t * G + g) * ld + m] = C;
            }
            F = F + drv[6];
            /* step_temp — oracle/fiveeq_oracle.py: step_temp() */
            S0 = S0 + em1_d0 * (S0 - q0 * F);
            S1 = S1 + em1_d1 * (S1 - q1 * F);
            if (T_traj) T_traj[(int64_t)t * ld + m] = S0 + S1;
        }
        for (int k = 0; k < off[G]; ++k) R[k * ld + m] = Rm[k];
        S[m] = S0;
        S[ld + m] = S1;
    }
    return 0;
}

/* emissions[0]*exp(-time) — U_FaIR/concentrations.py:5 of the reference; out[k][m] = e0[m]*exp(-time[k]) */
int oracle_hfc_conc(int64_t n, int64_t ld, int32_t n_time, const double *e0, const double *time, double *out) {
    if (!e0 || !time || !out || n < 1 || ld < n || n_time < 0) return -1;
    for (int32_t k = 0; k < n_time; ++k) {
        const double dec = exp(-time[k]);
        for (int64_t m = 0; m < n; ++m) out[(int64_t)k * ld + m] = e0[m] * dec;
    }
    return 0;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
