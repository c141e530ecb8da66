// Does the 32x32x16 fp16 MFMA sustain a higher rate than the 16x16x32 one under the power limit?  Same flops, same 64 x 64 x 32 wave tile, same 64 accumulator
// registers; register-fed (operands loaded once) and LDS-fed (4 + 4 ds_read_b128 per 64 x 64 x 32 step, the tap kernel's ratio).  Two 8-wave workgroups
// per CU, random operands, launches of ~100 us back to back for >= 1.5 s.   make -C tools/probes mfma_shape_probe && tools/probes/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef float float16v __attribute__((ext_vector_type(16)));

template <bool BIG, bool LDS>
__global__ void __launch_bounds__(512, 4) k(const half8 *src, float *sink, int iters) {
    __shared__ half8 lds[2048];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = src[i];
    __syncthreads();
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x + 64 * i) & 2047]; b[i] = src[(threadIdx.x * 3 + 64 * i + 17) & 2047]; }
    float s = 0.f;
    if constexpr (BIG) {
        float16v acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
            if constexpr (LDS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { a[i] = lds[(lane + 64 * i + it * 7) & 2047]; b[i] = lds[(lane + 64 * (i + 4) + it * 13) & 2047]; }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * ks + i], b[2 * ks + j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    } else {
        float4v acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            if constexpr (LDS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { a[i] = lds[(lane + 64 * i + it * 7) & 2047]; b[i] = lds[(lane + 64 * (i + 4) + it * 13) & 2047]; }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    }
    if (s == 123.456f) sink[threadIdx.x] = s;
}

// The LDS-fed 16x16x32 loop with the tap kernel's per-step structure: 4 weight-like + 2 position-like fragment reads per 16 MFMAs, optionally a workgroup
// barrier per step (what phase-locks the eight waves of a workgroup), at one or two 8-wave workgroups per CU.
template <bool BAR>
__global__ void __launch_bounds__(512, 4) kb(const half8 *src, float *sink, int iters) {
    __shared__ half8 lds[4096];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = src[i & 2047];
    __syncthreads();
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x + 64 * i) & 2047]; b[i] = src[(threadIdx.x * 3 + 64 * i + 17) & 2047]; }
    float4v acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if constexpr (BAR) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = lds[(lane + 64 * i + it * 7) & 4095];
#pragma unroll
        for (int i = 0; i < 2; ++i) b[i] = lds[(lane + 64 * (i + 4) + it * 13) & 4095];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) sink[threadIdx.x] = s;
}
template <bool BAR>
static double runb(const half8 *src, float *sink, int grid) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 170 * 512 / grid;
    int reps = 50;
    for (;;) {
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((kb<BAR>), dim3(grid), dim3(512), 0, 0, src, sink, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms * 1e-3 >= 1.0) break;
        reps *= 2;
    }
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((kb<BAR>), dim3(grid), dim3(512), 0, 0, src, sink, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return (double)grid * 8 * iters * (2.0 * 64 * 64 * 32) / (ms / reps * 1e-3) / 1e12;
}

template <bool BIG, bool LDS>
static double run(const half8 *src, float *sink) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 512, iters = 170;
    int reps = 50;
    for (;;) {
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<BIG, LDS>), dim3(grid), dim3(512), 0, 0, src, sink, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms * 1e-3 >= 1.5) break;
        reps *= 2;
    }
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<BIG, LDS>), dim3(grid), dim3(512), 0, 0, src, sink, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return (double)grid * 8 * iters * (2.0 * 64 * 64 * 32) / (ms / reps * 1e-3) / 1e12;
}

int main() {
    half8 *src; float *sink;
    CHECK(hipMalloc(&src, 4096 * sizeof(half8))); CHECK(hipMalloc(&sink, 4096));
    std::vector<_Float16> h(4096 * 8);
    srand(1);
    for (auto &v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
    CHECK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    const double r16 = run<false, false>(src, sink), r32 = run<true, false>(src, sink), l16 = run<false, true>(src, sink), l32 = run<true, true>(src, sink);
    const double r16b = run<false, false>(src, sink), r32b = run<true, false>(src, sink);
    const double f2 = runb<false>(src, sink, 512), b2 = runb<true>(src, sink, 512), f1 = runb<false>(src, sink, 256), b1 = runb<true>(src, sink, 256);
    printf("{\"tap_like_loop_tflops\": {\"two_wg_per_cu_free\": %.1f, \"two_wg_per_cu_barrier_per_step\": %.1f, \"one_wg_per_cu_free\": %.1f, \"one_wg_per_cu_barrier_per_step\": %.1f}}\n", f2, b2, f1, b1);
    printf("{\"register_fed_16x16x32_tflops\": [%.1f, %.1f], \"register_fed_32x32x16_tflops\": [%.1f, %.1f], \"lds_fed_16x16x32_tflops\": %.1f, \"lds_fed_32x32x16_tflops\": %.1f}\n",
           r16, r16b, r32, r32b, l16, l32);
    return 0;
}
