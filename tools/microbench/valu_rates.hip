// Microbenchmark: issue rate of the VALU instruction kinds the step kernels are made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates tools/microbench/valu_rates.hip && ./valu_rates
// Every kind is ONE instruction written as inline asm, so the compiler can neither fuse, nor drop, nor SLP-pack it
// (round 2's version used plain C++ and hipcc turned its eight scalar fp32 accumulators into four v_pk_* instructions:
// its "v_fma_f32 / v_mul_f32" rows were packed instructions counted as two).  Eight independent accumulators per lane
// cover the latency; W waves per SIMD (256-thread workgroups, 256 x W of them); reports cycles per wave-instruction per
// SIMD at 2.4 GHz.  "2 per lane" rows are packed: one instruction does two members' operations.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float float2v __attribute__((ext_vector_type(2)));

#define CHAIN(T, INIT, ASM, ...)                                     \
    {                                                                \
        T x[8];                                                      \
        for (int i = 0; i < 8; ++i) x[i] = INIT;                     \
        for (int it = 0; it < iters; ++it) {                         \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(x[i]) : __VA_ARGS__); \
        }                                                            \
        double s = 0;                                                \
        for (int i = 0; i < 8; ++i) s += sum(x[i]);                  \
        out[tid] = (float)s;                                         \
    }

__device__ inline double sum(float v) { return v; }
__device__ inline double sum(double v) { return v; }
__device__ inline double sum(float2v v) { return (double)v.x + v.y; }
__device__ inline double sum(int v) { return v; }

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const float2v a2{a, a}, b2{b, b};
    const double ad = a, bd = b;
    const int one = (iters & 1) ? 1 : -1;
    if constexpr (KIND == 0) CHAIN(float, a + i + tid, "v_fma_f32 %0, %0, %1, %2", "v"(a), "v"(b))
    if constexpr (KIND == 1) CHAIN(float2v, (float2v{a + i + tid, b + i}), "v_pk_fma_f32 %0, %0, %1, %2", "v"(a2), "v"(b2))
    if constexpr (KIND == 2) CHAIN(float, a + i + tid, "v_mul_f32 %0, %0, %1", "v"(a))
    if constexpr (KIND == 3) CHAIN(float2v, (float2v{a + i + tid, b + i}), "v_pk_mul_f32 %0, %0, %1", "v"(a2))
    if constexpr (KIND == 4) CHAIN(float, a + i + tid, "v_add_f32 %0, %0, %1", "v"(b))
    if constexpr (KIND == 5) CHAIN(float2v, (float2v{a + i + tid, b + i}), "v_pk_add_f32 %0, %0, %1", "v"(b2))
    if constexpr (KIND == 6) CHAIN(float, a + i + tid, "v_rcp_f32 %0, %0", "v"(a))
    if constexpr (KIND == 7) CHAIN(float, a + i + tid, "v_sqrt_f32 %0, %0", "v"(a))
    if constexpr (KIND == 8) CHAIN(float, a + i + tid, "v_exp_f32 %0, %0", "v"(a))
    if constexpr (KIND == 9) CHAIN(float, a + i + tid, "v_ldexp_f32 %0, %0, %1", "v"(one))
    if constexpr (KIND == 10) CHAIN(float, a + i + tid, "v_rndne_f32 %0, %0", "v"(a))
    if constexpr (KIND == 11) CHAIN(float, a + i + tid, "v_frexp_mant_f32 %0, %0", "v"(a))
    if constexpr (KIND == 12) CHAIN(float, a + i + tid, "v_min_f32 %0, %0, %1", "v"(b))
    if constexpr (KIND == 13) CHAIN(float, a + i + tid, "v_cvt_i32_f32 %0, %0", "v"(a))
    if constexpr (KIND == 14) CHAIN(float, a + i + tid, "v_cndmask_b32 %0, %0, %1, vcc", "v"(b))
    if constexpr (KIND == 15) CHAIN(float, a + i + tid, "v_cmp_gt_f32 vcc, %0, %1", "v"(b))
    if constexpr (KIND == 16) CHAIN(float, a + i + tid, "v_mov_b32 %0, %1", "v"(b))
    if constexpr (KIND == 17) CHAIN(float2v, (float2v{a + i + tid, b + i}), "v_pk_mov_b32 %0, %0, %1", "v"(b2))
    if constexpr (KIND == 20) CHAIN(double, ad + i + tid, "v_fma_f64 %0, %0, %1, %2", "v"(ad), "v"(bd))
    if constexpr (KIND == 21) CHAIN(double, ad + i + tid, "v_add_f64 %0, %0, %1", "v"(bd))
    if constexpr (KIND == 22) CHAIN(double, ad + i + tid, "v_mul_f64 %0, %0, %1", "v"(ad))
    if constexpr (KIND == 23) CHAIN(double, ad + i + tid, "v_rcp_f64 %0, %0", "v"(ad))
    if constexpr (KIND == 24) CHAIN(double, ad + i + tid, "v_ldexp_f64 %0, %0, %1", "v"(one))
    if constexpr (KIND == 25) CHAIN(double, ad + i + tid, "v_rndne_f64 %0, %0", "v"(ad))
    if constexpr (KIND == 26) CHAIN(double, ad + i + tid, "v_rsq_f64 %0, %0", "v"(ad))
    if constexpr (KIND == 27) CHAIN(double, ad + i + tid, "v_min_f64 %0, %0, %1", "v"(bd))
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
    const int cus = 256, iters = 2048;
    const int blocks = cus * waves_per_simd;            // 256 threads = 4 waves = one per SIMD
    float* out;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(out, 64, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(out, iters, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 8 * waves_per_simd;
    const double ns_per = ms * 1e6 / instr_per_simd;
    printf("%-30s waves/SIMD %d: %.3f ns per wave-instruction per SIMD = %5.2f cycles at 2.4 GHz\n", name, waves_per_simd,
           ns_per, ns_per * 2.4);
    (void)hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w);
        run<1>("v_pk_fma_f32 (2 per lane)", w);
        run<2>("v_mul_f32", w);
        run<3>("v_pk_mul_f32 (2 per lane)", w);
        run<4>("v_add_f32", w);
        run<5>("v_pk_add_f32 (2 per lane)", w);
        run<6>("v_rcp_f32", w);
        run<7>("v_sqrt_f32", w);
        run<8>("v_exp_f32", w);
        run<9>("v_ldexp_f32", w);
        run<10>("v_rndne_f32", w);
        run<11>("v_frexp_mant_f32", w);
        run<12>("v_min_f32", w);
        run<13>("v_cvt_i32_f32", w);
        run<14>("v_cndmask_b32", w);
        run<15>("v_cmp_gt_f32", w);
        run<16>("v_mov_b32", w);
        run<17>("v_pk_mov_b32", w);
        run<20>("v_fma_f64", w);
        run<21>("v_add_f64", w);
        run<22>("v_mul_f64", w);
        run<23>("v_rcp_f64", w);
        run<24>("v_ldexp_f64", w);
        run<25>("v_rndne_f64", w);
        run<26>("v_rsq_f64", w);
        run<27>("v_min_f64", w);
    }
    return 0;
}
