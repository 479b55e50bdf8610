// Microbenchmark: issue rate of the VALU instruction kinds the fused kernels are made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates tools/microbench/valu_rates.hip && ./valu_rates
// One wave per SIMD (256 threads per workgroup, 256 workgroups... x waves_per_simd), dependent chains of 8 independent
// accumulators per lane so latency is covered; reports cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    if constexpr (KIND == 0) {          // v_fma_f32
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = a + i + tid;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], a, b);
        float s = 0;
        for (int i = 0; i < 8; ++i) s += x[i];
        out[tid] = s;
    } else if constexpr (KIND == 1) {   // v_pk_fma_f32
        float2v x[8];
        for (int i = 0; i < 8; ++i) x[i] = float2v{a + i + tid, b + i};
        const float2v av{a, a}, bv{b, b};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
        float s = 0;
        for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
        out[tid] = s;
    } else if constexpr (KIND == 2) {   // v_fma_f64
        double x[8];
        for (int i = 0; i < 8; ++i) x[i] = a + i + tid;
        const double ad = a, bd = b;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], ad, bd);
        double s = 0;
        for (int i = 0; i < 8; ++i) s += x[i];
        out[tid] = (float)s;
    } else if constexpr (KIND == 3) {   // v_rcp_f64
        double x[8];
        for (int i = 0; i < 8; ++i) x[i] = a + i + tid;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_rcp(x[i]);
        double s = 0;
        for (int i = 0; i < 8; ++i) s += x[i];
        out[tid] = (float)s;
    } else if constexpr (KIND == 4) {   // v_ldexp_f64
        double x[8];
        for (int i = 0; i < 8; ++i) x[i] = a + i + tid;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_ldexp(x[i], (it & 1) ? 1 : -1);
        double s = 0;
        for (int i = 0; i < 8; ++i) s += x[i];
        out[tid] = (float)s;
    } else if constexpr (KIND == 5) {   // v_rndne_f64
        double x[8];
        for (int i = 0; i < 8; ++i) x[i] = a + i + tid;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_rint(x[i]) + 0.25;
        double s = 0;
        for (int i = 0; i < 8; ++i) s += x[i];
        out[tid] = (float)s;
    } else if constexpr (KIND == 6) {   // v_add_f64 (reference for KIND 5's extra add)
        double x[8];
        for (int i = 0; i < 8; ++i) x[i] = a + i + tid;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = x[i] + 0.25;
        double s = 0;
        for (int i = 0; i < 8; ++i) s += x[i];
        out[tid] = (float)s;
    } else if constexpr (KIND == 7) {   // v_mul_f32
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = a + i + tid;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = x[i] * a;
        float s = 0;
        for (int i = 0; i < 8; ++i) s += x[i];
        out[tid] = s;
    } else if constexpr (KIND == 8) {   // v_cndmask_b32 + v_cmp (select chain)
        float x[8];
        for (int i = 0; i < 8; ++i) x[i] = a + i + tid;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = x[i] > b ? x[i] : a;
        float s = 0;
        for (int i = 0; i < 8; ++i) s += x[i];
        out[tid] = s;
    }
}

template <int KIND>
void run(const char* name, int instr_per_iter_per_acc, int waves_per_simd) {
    const int cus = 256, iters = 4096;
    const int blocks = cus * waves_per_simd;            // 256 threads = 4 waves = one per SIMD
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(out, 64, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 8 * instr_per_iter_per_acc * waves_per_simd;
    const double ns_per = ms * 1e6 / instr_per_simd;
    printf("%-34s waves/SIMD %d: %.3f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz\n", name, waves_per_simd, ns_per,
           ns_per * 2.4);
    hipFree(out);
}

int main() {
    for (int w : {1, 4}) {
        run<0>("v_fma_f32", 1, w);
        run<1>("v_pk_fma_f32 (2 FMA per lane)", 1, w);
        run<7>("v_mul_f32", 1, w);
        run<8>("v_cmp + v_cndmask_b32", 2, w);
        run<2>("v_fma_f64", 1, w);
        run<6>("v_add_f64", 1, w);
        run<3>("v_rcp_f64", 1, w);
        run<4>("v_ldexp_f64", 1, w);
        run<5>("v_rndne_f64 + v_add_f64", 2, w);
    }
    return 0;
}
