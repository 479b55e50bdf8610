// launch_cost.hip — host time of one kernel launch with the step kernel's argument shape (a 496-byte struct by value + 19
// pointers / scalars), three ways: hipLaunchKernelGGL (per-argument marshalling through the kernel's metadata),
// hipModuleLaunchKernel with a pre-packed argument buffer (HIP_LAUNCH_PARAM_BUFFER_POINTER: one copy), and a captured graph of
// 20 such launches.  The kernel does nothing; the stream is drained between bursts so the queue never fills.
//   hipcc --offload-arch=gfx950 -O2 launch_cost.hip -o launch_cost && ./launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

struct Model { double v[62]; };   // 496 bytes, like KModel<double>

__global__ __launch_bounds__(64) void like_step(const Model km, const double* drive, const int n_steps, const int t, const long n,
                                                const long ld, const double* r, const double* q, double* R, double* S,
                                                double* C, double* T, const int n_rows, double* stats, unsigned short* ring,
                                                const int ring_rows, const double lo, const double inv_w, const int n_bins) {
    if (n < 0) S[0] = km.v[t & 63] + drive[0] + r[0] + q[0] + R[0] + C[0] + T[0] + stats[0] + ring[0] + lo + inv_w + n_steps + ld +
                      n_rows + ring_rows + n_bins;
}

struct Packed {      // the kernarg segment as the compiler lays it out: natural alignment of every argument
    Model km; const double* drive; int n_steps; int t; long n; long ld; const double* r; const double* q; double* R; double* S;
    double* C; double* T; int n_rows; int pad0; double* stats; unsigned short* ring; int ring_rows; int pad1; double lo; double inv_w;
    int n_bins; int pad2;
};

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    double* buf;
    CK(hipMalloc(&buf, 1 << 20));
    Model km;
    memset(&km, 0, sizeof km);
    const int burst = 40, reps = 200;
    const long n = 15625 * 64;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    double best[3] = {1e9, 1e9, 1e9};
    hipFunction_t fn;
    CK(hipGetFuncBySymbol(&fn, reinterpret_cast<const void*>(like_step)));
    Packed pk;
    memset(&pk, 0, sizeof pk);
    pk.km = km; pk.drive = buf; pk.n_steps = 750; pk.n = n; pk.ld = n; pk.r = buf; pk.q = buf; pk.R = buf; pk.S = buf; pk.C = buf; pk.T = buf;
    pk.n_rows = 750; pk.stats = buf; pk.ring = (unsigned short*)buf; pk.ring_rows = 1; pk.lo = 0; pk.inv_w = 1; pk.n_bins = 1;
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < burst; ++i)
        hipLaunchKernelGGL(like_step, dim3(15625), dim3(64), 0, st, km, buf, 750, i, n, n, buf, buf, buf, buf, buf, buf, 750, buf,
                           (unsigned short*)buf, 1, 0.0, 1.0, 1);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int r = 0; r < reps; ++r) {
        CK(hipStreamSynchronize(st));
        auto t0 = now();
        for (int i = 0; i < burst; ++i)
            hipLaunchKernelGGL(like_step, dim3(15625), dim3(64), 0, st, km, buf, 750, i, n, n, buf, buf, buf, buf, buf, buf, 750, buf,
                               (unsigned short*)buf, 1, 0.0, 1.0, 1);
        auto t1 = now();
        best[0] = std::min(best[0], us(t0, t1) / burst);
        CK(hipStreamSynchronize(st));
        t0 = now();
        for (int i = 0; i < burst; ++i) {
            pk.t = i;
            size_t sz = sizeof pk;
            void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pk, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            if (hipModuleLaunchKernel(fn, 15625, 1, 1, 64, 1, 1, 0, st, nullptr, extra) != hipSuccess) return 2;
        }
        t1 = now();
        best[1] = std::min(best[1], us(t0, t1) / burst);
        CK(hipStreamSynchronize(st));
        t0 = now();
        CK(hipGraphLaunch(exec, st));
        t1 = now();
        best[2] = std::min(best[2], us(t0, t1) / burst);
    }
    CK(hipStreamSynchronize(st));
    printf("host time per launch, step-kernel argument shape (%zu-byte kernarg), bursts of %d on a drained stream, best of %d:\n",
           sizeof(Packed), burst, reps);
    printf("  hipLaunchKernelGGL                              %6.2f us\n", best[0]);
    printf("  hipModuleLaunchKernel, pre-packed buffer        %6.2f us\n", best[1]);
    printf("  hipGraphLaunch of a %d-launch graph, per launch %6.2f us\n", burst, best[2]);
    return 0;
}
