// host_example.cpp — the C ABI (include/fiveeq.h) driven from a plain C++ host: no Python, no torch.
//
//   hipcc --offload-arch=gfx950 -O2 -I include example/host_example.cpp -L fiveeqscm_amd/csrc -lfiveeq_hip \
//         -Wl,-rpath,$PWD/fiveeqscm_amd/csrc -o example/host_example        (or: make -C example)
//   ./example/host_example [n_members] [n_steps]
//
// What a C/C++ caller does: fill `fiveeq_model` (shared parameters), put the per-member parameter rows and the state rows
// in device memory, build the drive table, and call fiveeq_run_f64 — one kernel launch per timestep, enqueued from C.
// The member parameters are a shard-computable Latin hypercube drawn ON THE DEVICE by the library (fiveeq_lhs_rows_f64);
// the perturbation rule (r0 x0.8..1.2 etc., TCR/ECS -> q) is applied here on the host for brevity.
// Prints per-step ensemble statistics of T at a few years; tests/test_host_example.py compares them with the Python engine.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fiveeq.h"

#define CHECK_HIP(x)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)
#define CHECK_FE(x)                                                                        \
    do {                                                                                   \
        int rc_ = (x);                                                                     \
        if (rc_ != FIVEEQ_OK) {                                                            \
            std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, fiveeq_last_error());          \
            return 3;                                                                      \
        }                                                                                  \
    } while (0)

static double sigma(double x) { return 1.0 / (1.0 + std::exp(-x)); }

// closed-form alpha constants (names from .coveragerc:15-16 of the reference)
static double g_1(const double* a, const double* tau, int n) {
    double s = 0;
    for (int i = 0; i < n; ++i) {
        const double x = 100.0 / tau[i];
        s += a[i] * tau[i] * (x < 0.05 ? x * x * (0.5 - x / 3.0 + x * x / 8.0) : 1.0 - (1.0 + x) * std::exp(-x));
    }
    return s;
}
static double g_0(const double* a, const double* tau, int n) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += a[i] * tau[i] * (-std::expm1(-100.0 / tau[i]));
    return std::exp(-s / g_1(a, tau, n));
}

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? std::atoll(argv[1]) : 100000;
    const int n_steps = argc > 2 ? std::atoi(argv[2]) : 750;
    if (fiveeq_abi_version() != FIVEEQ_ABI_VERSION || fiveeq_sizeof_model() != (int64_t)sizeof(fiveeq_model)) {
        std::fprintf(stderr, "library / header mismatch\n");
        return 1;
    }
    // ---- shared model: CO2 only, Millar-2017 values (fiveeqscm_amd.params.default_params("co2")) ----
    fiveeq_model m = {};
    m.n_gas = 1;
    m.dt = 1.0;
    m.iirf_max = 97.0;
    m.d[0] = 239.0;
    m.d[1] = 4.1;
    fiveeq_gas& g = m.gas[0];
    const double a[4] = {0.2173, 0.2240, 0.2824, 0.2763}, tau[4] = {1.0e6, 394.4, 36.54, 4.304};
    for (int i = 0; i < 4; ++i) g.a[i] = a[i], g.tau[i] = tau[i];
    g.n_pools = 4;
    g.g0 = g_0(a, tau, 4);
    g.g1 = g_1(a, tau, 4);
    g.ra = 0.0;
    g.C0 = 278.0;
    g.emis2conc = 1.0 / 2.123;
    const double F2x = 3.74;
    g.f[0] = F2x / std::log(2.0);
    const int32_t pools[1] = {4};
    if (!fiveeq_layout_supported(1, pools)) return 1;

    // ---- drive table: RCP-like CO2 emissions, every step stored at row t ----
    std::vector<double> drive((size_t)n_steps * FIVEEQ_DRIVE_STRIDE, 0.0);
    double cum = 0.0;
    for (int t = 0; t < n_steps; ++t) {
        double E = 12.0 * sigma((t - 230.0) / 25.0) * (1.0 - 1.1 * sigma((t - 330.0) / 20.0));
        if (E < -1.0) E = -1.0;
        drive[(size_t)t * 8 + 0] = E;
        drive[(size_t)t * 8 + 3] = cum;          // cumulative emissions BEFORE the step
        drive[(size_t)t * 8 + 7] = t;            // output row
        cum += E * m.dt;
    }

    // ---- member parameters: 5 Latin-hypercube dimensions drawn on the device, perturbation rule on the host ----
    double *d_u = nullptr, *d_r = nullptr, *d_q = nullptr, *d_R = nullptr, *d_S = nullptr, *d_T = nullptr, *d_drive = nullptr,
           *d_stats = nullptr;
    CHECK_HIP(hipMalloc(&d_u, 5 * N * sizeof(double)));
    CHECK_FE(fiveeq_lhs_rows_f64(20261003ULL, N, 0, N, 0, 5, N, d_u, nullptr));
    std::vector<double> u(5 * N), r(3 * N), q(2 * N);
    CHECK_HIP(hipMemcpy(u.data(), d_u, u.size() * sizeof(double), hipMemcpyDeviceToHost));
    const double k1 = 1.0 - (m.d[0] / 70.0) * (-std::expm1(-70.0 / m.d[0])), k2 = 1.0 - (m.d[1] / 70.0) * (-std::expm1(-70.0 / m.d[1]));
    const double inv_den = 1.0 / (F2x * (k1 - k2));
    for (int64_t i = 0; i < N; ++i) {
        r[0 * N + i] = ((1.2 - 0.8) * u[0 * N + i] + 0.8) * 32.4;   // r0: the rule of params.sample_ensemble_shard
        r[1 * N + i] = ((1.5 - 0.5) * u[1 * N + i] + 0.5) * 0.019;  // rC
        r[2 * N + i] = ((1.5 - 0.5) * u[2 * N + i] + 0.5) * 4.165;  // rT
        double tcr = 1.5 * u[3 * N + i] + 1.0, ecs = 3.0 * u[4 * N + i] + 1.5;
        if (ecs < tcr) std::swap(tcr, ecs);
        if (ecs < 1.1 * tcr) ecs = 1.1 * tcr;
        q[0 * N + i] = (tcr - ecs * k2) * inv_den;
        q[1 * N + i] = (ecs * k1 - tcr) * inv_den;
    }
    const int64_t W = fiveeq_stats_waves(N);
    CHECK_HIP(hipMalloc(&d_r, r.size() * sizeof(double)));
    CHECK_HIP(hipMalloc(&d_q, q.size() * sizeof(double)));
    CHECK_HIP(hipMalloc(&d_R, 4 * N * sizeof(double)));
    CHECK_HIP(hipMalloc(&d_S, 2 * N * sizeof(double)));
    CHECK_HIP(hipMalloc(&d_T, (size_t)n_steps * N * sizeof(double)));
    CHECK_HIP(hipMalloc(&d_drive, drive.size() * sizeof(double)));
    CHECK_HIP(hipMalloc(&d_stats, (size_t)W * n_steps * 4 * sizeof(double)));
    CHECK_HIP(hipMemcpy(d_r, r.data(), r.size() * sizeof(double), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_q, q.data(), q.size() * sizeof(double), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_drive, drive.data(), drive.size() * sizeof(double), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemset(d_R, 0, 4 * N * sizeof(double)));
    CHECK_HIP(hipMemset(d_S, 0, 2 * N * sizeof(double)));

    // ---- the run: one kernel launch per timestep, enqueued from C; then the same through the K-steps form ----
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CHECK_HIP(hipEventCreate(&e0));
    CHECK_HIP(hipEventCreate(&e1));
    CHECK_HIP(hipEventRecord(e0, st));
    CHECK_FE(fiveeq_run_f64(&m, N, N, d_drive, n_steps, 0, n_steps, d_r, d_q, d_R, d_S, nullptr, d_T, n_steps, d_stats, st));
    CHECK_HIP(hipEventRecord(e1, st));
    CHECK_HIP(hipStreamSynchronize(st));
    float ms = 0;
    CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));

    std::vector<double> stats((size_t)W * n_steps * 4), Tlast(N);
    CHECK_HIP(hipMemcpy(stats.data(), d_stats, stats.size() * sizeof(double), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(Tlast.data(), d_T + (size_t)(n_steps - 1) * N, N * sizeof(double), hipMemcpyDeviceToHost));
    std::printf("host_example: %lld members x %d steps, CO2 only, fp64, per-step kernel: %.3f ms = %.3e member-timesteps/s\n",
                (long long)N, n_steps, ms, (double)N * n_steps / (ms * 1e-3));
    const int years[3] = {n_steps / 3, 2 * n_steps / 3, n_steps - 1};
    for (int t : years) {
        double s1 = 0, mn = 1e300, mx = -1e300;
        for (int64_t w = 0; w < W; ++w) {
            const double* rec = &stats[((size_t)w * n_steps + t) * 4];
            s1 += rec[0];
            mn = std::fmin(mn, rec[2]);
            mx = std::fmax(mx, rec[3]);
        }
        std::printf("step %d: T mean %.15g min %.15g max %.15g\n", t, s1 / N, mn, mx);
    }
    double direct = 0;
    for (int64_t i = 0; i < N; ++i) direct += Tlast[i];
    std::printf("step %d: T mean from the stored row %.15g\n", n_steps - 1, direct / N);
    for (double* ptr : {d_u, d_r, d_q, d_R, d_S, d_T, d_drive, d_stats}) (void)hipFree(ptr);
    return 0;
}
