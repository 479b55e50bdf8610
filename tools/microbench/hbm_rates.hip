// hbm_rates.hip — what this box's HBM does for plain streaming kernels, beyond the Infinity Cache (2 x 2 GiB buffers): the
// ceiling `roofline.hbm_resident_frac` is held against.  Shapes: copy (1 read : 1 write), read-only, write-only, and the per-step
// kernel's mix (19 rows read : 12 rows written, row-strided like its struct-of-arrays state); access width 8 / 16 B per lane;
// default and non-temporal policy; grids from one workgroup per CU to one per 4 KiB.
//   hipcc --offload-arch=gfx950 -O3 hbm_rates.hip -o hbm_rates && ./hbm_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

typedef double double2v __attribute__((ext_vector_type(2)));

template <typename V, bool NT>
__device__ __forceinline__ V ld(const V* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <typename V, bool NT>
__device__ __forceinline__ void st(V* p, V v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// each workgroup walks contiguous tiles of UNROLL x 256 elements, UNROLL loads in flight per lane
template <typename V, int UNROLL, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_k(const int64_t n, const V* __restrict__ src, V* __restrict__ dst) {
    const int64_t tile = (int64_t)UNROLL * 256;
    for (int64_t base = (int64_t)blockIdx.x * tile; base + tile <= n; base += (int64_t)gridDim.x * tile) {
        V v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = ld<V, NTL>(src + base + u * 256 + threadIdx.x);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) st<V, NTS>(dst + base + u * 256 + threadIdx.x, v[u]);
    }
}

template <typename V, int UNROLL, bool NTL>
__global__ __launch_bounds__(256) void read_k(const int64_t n, const V* __restrict__ src, double* __restrict__ out) {
    const int64_t tile = (int64_t)UNROLL * 256;
    double acc = 0.0;
    for (int64_t base = (int64_t)blockIdx.x * tile; base + tile <= n; base += (int64_t)gridDim.x * tile) {
        V v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = ld<V, NTL>(src + base + u * 256 + threadIdx.x);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if constexpr (sizeof(V) == 16) acc += v[u].x + v[u].y;
            else acc += v[u];
        }
    }
    if (acc == 12345.678) out[0] = acc;
}

template <typename V, int UNROLL, bool NTS>
__global__ __launch_bounds__(256) void write_k(const int64_t n, V* __restrict__ dst) {
    const int64_t tile = (int64_t)UNROLL * 256;
    V z;
    if constexpr (sizeof(V) == 16) z = V{1.0, 2.0};
    else z = 1.0;
    for (int64_t base = (int64_t)blockIdx.x * tile; base + tile <= n; base += (int64_t)gridDim.x * tile) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) st<V, NTS>(dst + base + u * 256 + threadIdx.x, z);
    }
}

// the per-step kernel's shape: one member per lane, 19 rows read, 12 rows written (8 of them in place), one wave per workgroup
template <bool NTL, bool NTS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void step_like_k(const int64_t n, const double* __restrict__ in, double* __restrict__ state,
                                                     double* __restrict__ out) {
    const int64_t m = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (m >= n) return;
    double p[11], s[8];
#pragma unroll
    for (int k = 0; k < 11; ++k) p[k] = ld<double, NTL>(in + k * n + m);
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = ld<double, NTL>(state + k * n + m);
    double c = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = s[k] * 0.999 + p[k], c += s[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) st<double, NTS>(state + k * n + m, s[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) st<double, NTS>(out + k * n + m, c + p[8 + (k % 3)]);
}

// the same traffic from a TILE-MAJOR layout: the 19 input rows of a wave's 64 members are one contiguous 9.5 KiB block
// [tile][row][64] (state rows first, written back in place); the 4 output rows stay member-major
template <bool NTL, bool NTS>
__global__ __launch_bounds__(64) void step_like_tiled_k(const int64_t n, double* __restrict__ tiles, double* __restrict__ out) {
    const int64_t m = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (m >= n) return;
    double* t = tiles + (int64_t)blockIdx.x * 19 * 64 + threadIdx.x;
    double p[11], s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = ld<double, NTL>(t + k * 64);
#pragma unroll
    for (int k = 0; k < 11; ++k) p[k] = ld<double, NTL>(t + (8 + k) * 64);
    double c = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = s[k] * 0.999 + p[k], c += s[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) st<double, NTS>(t + k * 64, s[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) st<double, NTS>(out + k * n + m, c + p[8 + (k % 3)]);
}

static hipEvent_t e0, e1;
template <typename F>
double time_ms(F launch, int reps) {
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int64_t bytes = (int64_t)2 << 30;                 // 2 GiB per buffer: 8x the Infinity Cache
    const int64_t n8 = bytes / 8, n16 = bytes / 16;
    double *a, *b;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes));
    CK(hipMemset(b, 0, bytes));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# %s, %d CUs; 2 GiB buffers; GB/s of bytes read + written; best of the grids listed\n", prop.name, cus);
    const int reps = 6;
    const std::vector<int> per_cu = {1, 2, 4, 8, 16, 32, 0};      // workgroups per CU; 0 = one workgroup per tile
    auto grids = [&](int64_t n, int64_t tile, auto fn, const char* name, double bytes_moved) {
        double best = 0.0;
        int best_g = 0;
        printf("%-44s", name);
        for (int k : per_cu) {
            const int64_t g = k ? (int64_t)k * cus : n / tile;
            const double ms = time_ms([&] { fn((int)g); }, reps);
            const double gbs = bytes_moved / (ms * 1e-3) / 1e9;
            printf(" %7.0f", gbs);
            if (gbs > best) best = gbs, best_g = k;
        }
        printf("   best %.0f (%d per CU)\n", best, best_g);
    };
    printf("# %-42s", "workgroups per CU:");
    for (int k : per_cu) printf(" %7d", k);
    printf("\n");
#define COPY(V, n, U, NL, NS, name) grids(n, (int64_t)U * 256, [&](int g) { hipLaunchKernelGGL((copy_k<V, U, NL, NS>), dim3(g), dim3(256), 0, 0, n, (const V*)a, (V*)b); }, name, 2.0 * bytes)
    COPY(double, n8, 4, false, false, "copy 8 B/lane x4");
    COPY(double2v, n16, 4, false, false, "copy 16 B/lane x4");
    COPY(double2v, n16, 8, false, false, "copy 16 B/lane x8");
    COPY(double2v, n16, 2, false, false, "copy 16 B/lane x2");
    COPY(double2v, n16, 4, true, false, "copy 16 B/lane x4, nt loads");
    COPY(double2v, n16, 4, false, true, "copy 16 B/lane x4, nt stores");
    COPY(double2v, n16, 4, true, true, "copy 16 B/lane x4, nt both");
    COPY(double, n8, 4, true, true, "copy 8 B/lane x4, nt both");
#define READ(V, n, U, NL, name) grids(n, (int64_t)U * 256, [&](int g) { hipLaunchKernelGGL((read_k<V, U, NL>), dim3(g), dim3(256), 0, 0, n, (const V*)a, b); }, name, 1.0 * bytes)
    READ(double, n8, 4, false, "read 8 B/lane x4");
    READ(double2v, n16, 4, false, "read 16 B/lane x4");
    READ(double2v, n16, 4, true, "read 16 B/lane x4, nt");
#define WRITE(V, n, U, NS, name) grids(n, (int64_t)U * 256, [&](int g) { hipLaunchKernelGGL((write_k<V, U, NS>), dim3(g), dim3(256), 0, 0, n, (V*)b); }, name, 1.0 * bytes)
    WRITE(double, n8, 4, false, "write 8 B/lane x4");
    WRITE(double2v, n16, 4, false, "write 16 B/lane x4");
    WRITE(double2v, n16, 4, true, "write 16 B/lane x4, nt");
    // the step kernel's mix at 8M members: 11 parameter rows (704 MB) + 8 state rows in place (512 MB) + 4 output rows (256 MB)
    const int64_t nm = 8000000;
    double* out = b + 8 * nm;
    const double moved = (19.0 + 12.0) * 8.0 * nm;
    printf("# the per-step kernel's mix, 8M members (19 rows read, 12 written; 248 B per member), one launch:\n");
#define STEPLIKE(NL, NS, B, name) { const double ms = time_ms([&] { hipLaunchKernelGGL((step_like_k<NL, NS, B>), dim3((nm + B - 1) / B), dim3(B), 0, 0, nm, (const double*)a, b, out); }, reps); \
        printf("%-44s %7.0f GB/s  (%.1f us)\n", name, moved / (ms * 1e-3) / 1e9, ms * 1e3); }
    STEPLIKE(false, false, 64, "step-like, 64-thread workgroups");
    STEPLIKE(false, false, 256, "step-like, 256-thread workgroups");
    STEPLIKE(true, false, 64, "step-like, nt loads");
    STEPLIKE(false, true, 64, "step-like, nt stores");
    STEPLIKE(true, true, 64, "step-like, nt both");
    STEPLIKE(true, true, 256, "step-like, nt both, 256 threads");
#define STEPTILED(NL, NS, name) { const double ms = time_ms([&] { hipLaunchKernelGGL((step_like_tiled_k<NL, NS>), dim3((nm + 63) / 64), dim3(64), 0, 0, nm, a, b); }, reps); \
        printf("%-44s %7.0f GB/s  (%.1f us)\n", name, moved / (ms * 1e-3) / 1e9, ms * 1e3); }
    STEPTILED(false, false, "step-like, tile-major inputs");
    STEPTILED(true, true, "step-like, tile-major inputs, nt both");
    STEPTILED(false, true, "step-like, tile-major inputs, nt stores");
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    return 0;
}
