// Accuracy of the hardware base-2 logarithm (v_log_f32) over the range the CO2 forcing feeds it (C/C0 in [0.5, 16]) and
// close to 1, where ln(x) -> 0 and a table-based log2 may lose RELATIVE accuracy.  Prints ulp(float) errors of
//   (a) v_log_f32(x) against log2(x), (b) ln2 * v_log_f32(x) (hi/lo product) against ln(x).
//   hipcc --offload-arch=gfx950 -O2 -o hw_log_accuracy hw_log_accuracy.hip && ./hw_log_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* y2, float* ye, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float t = __builtin_amdgcn_logf(x[i]);
    y2[i] = t;
    const float hi = 0x1.62e430p-1f, lo = -0x1.05c610p-29f;      // ln2 = hi + lo
    float p = t * hi;
    float e = __builtin_fmaf(t, hi, -p);
    ye[i] = p + __builtin_fmaf(t, lo, e);
}
static double ulp_of(double v) { float f = (float)fabs(v); return (double)(nextafterf(f, INFINITY) - f); }
int main() {
    struct R { const char* name; double a, b; } ranges[] = {{"[0.5, 16]", 0.5, 16}, {"[1, 1.1]", 1.0, 1.1}, {"[1, 1.001]", 1.0, 1.001},
                                                            {"[1, 1+1e-5]", 1.0, 1.00001}, {"[0.99, 1]", 0.99, 1.0}, {"[1.1, 8]", 1.1, 8}};
    const int n = 1 << 22;
    float *dx, *d2, *de;
    hipMalloc(&dx, n * 4); hipMalloc(&d2, n * 4); hipMalloc(&de, n * 4);
    std::vector<float> x(n), y2(n), ye(n);
    for (auto& r : ranges) {
        for (int i = 0; i < n; ++i) x[i] = (float)(r.a + (r.b - r.a) * ((i + 0.5) / n));
        hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d2, de, n);
        hipMemcpy(y2.data(), d2, n * 4, hipMemcpyDeviceToHost); hipMemcpy(ye.data(), de, n * 4, hipMemcpyDeviceToHost);
        double m2 = 0, me = 0, a2 = 0, ae = 0; int cnt = 0;
        for (int i = 0; i < n; ++i) {
            double w2 = log2((double)x[i]), we = log((double)x[i]);
            if (w2 == 0) continue;
            double u2 = fabs(y2[i] - w2) / ulp_of(w2), ue = fabs(ye[i] - we) / ulp_of(we);
            m2 = fmax(m2, u2); me = fmax(me, ue); a2 += u2; ae += ue; ++cnt;
        }
        printf("x in %-12s  v_log_f32 vs log2: max %.2f ulp, mean %.3f   |  ln2*v_log_f32 vs ln: max %.2f ulp, mean %.3f\n", r.name, m2, a2 / cnt, me, ae / cnt);
    }
    return 0;
}
