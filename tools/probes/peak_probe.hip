// Probe (not product): the ceilings the roofline fractions are read against, measured on the chip bench.py runs on.
//   1. fp16 MFMA (v_mfma_f32_16x16x32_f16, the instruction of the conv kernels) on RANDOM operands held in registers:
//      8 waves per workgroup, two workgroups per CU (the residency of the tap-reuse kernel), 16 independent accumulators per
//      wave, launches of ~100-150 us repeated back to back for >= 1 s -- the clock the chip HOLDS under matrix load, not the
//      2.4 GHz it starts from (MI355X_MICROARCH.md, DVFS give-back).  Also on zero operands (the power-free upper bound).
//   2. the same loop with every operand re-read from LDS by ds_read_b128 (one A and one B fragment per four MFMAs, the tap
//      kernel's ratio): what an LDS-fed inner loop can reach.
//   3. fp32 MFMA (v_mfma_f32_16x16x4_f32), random operands.
//   4. HBM: float4 copy of 1 GiB (read + write bytes / time).
// Prints ONE JSON line; in-kernel clock = s_memtime ticks / s_memrealtime ticks x 100 MHz (median over workgroups).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool LDS>
__global__ void __launch_bounds__(512, 4) mfma_f16_kernel(const half8 *src, float *sink, unsigned long long *clk, int iters) {
    __shared__ half8 lds[LDS ? 2048 : 1];
    const int lane = threadIdx.x & 63;
    half8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x * 4 + i) & 4095]; b[i] = src[(threadIdx.x * 4 + i + 2048) & 4095]; }
    if (LDS) {
        for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = src[i];
        __syncthreads();
    }
    float4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (LDS) {      // 4 + 4 fragment reads per 16 MFMAs, rows chosen per lane and iteration (conflict-free: consecutive 16-byte rows)
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = lds[(lane + 64 * i + it * 7) & 2047]; b[i] = lds[(lane + 64 * (i + 4) + it * 13) & 2047]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) sink[threadIdx.x] = s;       // keeps the accumulators alive
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

__global__ void __launch_bounds__(256, 4) mfma_f32_kernel(const float *src, float *sink, int iters) {
    float a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x * 4 + i) & 4095]; b[i] = src[(threadIdx.x * 4 + i + 1024) & 4095]; }
    float4v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) sink[threadIdx.x] = s;
}

// each workgroup owns a contiguous 64 KiB span per pass; four independent 16-byte loads per lane in flight
__global__ void __launch_bounds__(256) copy_kernel(const float4v *__restrict__ in, float4v *__restrict__ out, size_t n) {
    const size_t span = 256 * 4;
    for (size_t base = (size_t)blockIdx.x * span; base + span <= n; base += (size_t)gridDim.x * span) {
        float4v v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(in + base + k * 256 + threadIdx.x);
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(v[k], out + base + k * 256 + threadIdx.x);
    }
}
__global__ void __launch_bounds__(256) read_kernel(const float4v *__restrict__ in, float *sink, size_t n) {
    const size_t span = 256 * 8;
    float acc = 0.f;
    for (size_t base = (size_t)blockIdx.x * span; base + span <= n; base += (size_t)gridDim.x * span) {
        float4v v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = in[base + k * 256 + threadIdx.x];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].w;
    }
    if (acc == 123.456f) sink[threadIdx.x] = acc;
}

static double elapsed_ms(hipEvent_t a, hipEvent_t b) { float ms = 0.f; CHECK(hipEventElapsedTime(&ms, a, b)); return ms; }

// LDS-fed loop with a TM x TP fragment tile per wave (TM + TP ds_read_b128 per TM * TP MFMAs) and WAVES waves per workgroup:
// what register tile / occupancy combination an LDS-fed conv loop should aim for.
template <int TM, int TP, int WAVES, int OCC>
__global__ void __launch_bounds__(WAVES * 64, OCC) mfma_f16_tile_kernel(const half8 *src, float *sink, int iters) {
    __shared__ half8 lds[2048];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += WAVES * 64) lds[i] = src[i];
    __syncthreads();
    float4v acc[TM][TP];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) acc[i][j] = float4v{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        half8 a[TM], b[TP];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = lds[(lane + 64 * i + it * 7) & 2047];
#pragma unroll
        for (int j = 0; j < TP; ++j) b[j] = lds[(lane + 64 * (j + TM) + it * 13) & 2047];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TP; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TP; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 123.456f) sink[threadIdx.x] = s;
}

template <int TM, int TP, int WAVES, int OCC>
static double run_tile(const half8 *src, float *sink, int grid, double min_seconds) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 170 * 16 / (TM * TP) * 16 / WAVES * 512 / grid;     // ~ the same flops per launch as the 4 x 4 / 8-wave / 512-workgroup run
    int reps = 50;
    for (;;) {
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mfma_f16_tile_kernel<TM, TP, WAVES, OCC>), dim3(grid), dim3(WAVES * 64), 0, 0, src, sink, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        if (elapsed_ms(e0, e1) * 1e-3 >= min_seconds) break;
        reps *= 2;
    }
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((mfma_f16_tile_kernel<TM, TP, WAVES, OCC>), dim3(grid), dim3(WAVES * 64), 0, 0, src, sink, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    const double ms = elapsed_ms(e0, e1) / reps;
    return (double)grid * WAVES * iters * (double)(TM * TP) * (2.0 * 16 * 16 * 32) / (ms * 1e-3) / 1e12;
}


template <bool LDS>
static void run_f16(const half8 *src, float *sink, unsigned long long *clk, int grid, int iters, double min_seconds, double &tflops, double &ghz, double &us) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // warm: run back to back until the clock has settled, then time `reps` launches
    int reps = 50;
    for (;;) {
        CHECK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_f16_kernel<LDS>, dim3(grid), dim3(512), 0, 0, src, sink, clk, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        if (elapsed_ms(e0, e1) * 1e-3 >= min_seconds) break;
        reps *= 2;
    }
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_f16_kernel<LDS>, dim3(grid), dim3(512), 0, 0, src, sink, clk, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    const double ms = elapsed_ms(e0, e1) / reps;
    us = ms * 1e3;
    const double flops = (double)grid * 8 /* waves */ * iters * 16.0 * (2.0 * 16 * 16 * 32);
    tflops = flops / (ms * 1e-3) / 1e12;
    std::vector<unsigned long long> h(2 * grid);
    CHECK(hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
    std::vector<double> g;
    for (int i = 0; i < grid; ++i) if (h[2 * i + 1]) g.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
    std::sort(g.begin(), g.end());
    ghz = g.empty() ? 0.0 : g[g.size() / 2];
}

int main() {
    const int grid = 512;           // two 8-wave workgroups per CU on 256 CUs
    half8 *src; float *sink; unsigned long long *clk;
    CHECK(hipMalloc(&src, 4096 * sizeof(half8))); CHECK(hipMalloc(&sink, 4096)); CHECK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * grid));
    std::vector<_Float16> h(4096 * 8);
    srand(1);
    for (auto &v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
    CHECK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    // iters: ~120 us per launch at ~1.5 PF: 512 WG * 8 waves * iters * 16 MFMA * 16384 flop = 1.07e9 * iters flop -> iters = 170
    double tf_rand, ghz_rand, us_rand, tf_lds, ghz_lds, us_lds, tf_zero, ghz_zero, us_zero;
    run_f16<false>(src, sink, clk, grid, 170, 1.5, tf_rand, ghz_rand, us_rand);
    run_f16<true>(src, sink, clk, grid, 170, 1.5, tf_lds, ghz_lds, us_lds);
    // register tile / occupancy variants of the LDS-fed loop (random operands)
    const double t44_16 = run_tile<4, 4, 8, 4>(src, sink, 512, 1.0);      // 4 x 4 tile, 16 waves per CU (the tap kernel's shape)
    const double t84_8 = run_tile<8, 4, 8, 2>(src, sink, 256, 1.0);       // 8 x 4 tile, one 8-wave workgroup per CU
    const double t84_2x4 = run_tile<8, 4, 4, 2>(src, sink, 512, 1.0);     // 8 x 4 tile, two 4-wave workgroups per CU
    const double t48_2x4 = run_tile<4, 8, 4, 2>(src, sink, 512, 1.0);     // 4 x 8
    const double t47_8 = run_tile<4, 7, 8, 2>(src, sink, 256, 1.0);       // 4 x 7 (the 19 x 19 tile)
    CHECK(hipMemset(src, 0, 4096 * sizeof(half8)));
    run_f16<false>(src, sink, clk, grid, 170, 1.5, tf_zero, ghz_zero, us_zero);
    // fp32 MFMA
    float *fsrc; CHECK(hipMalloc(&fsrc, 4096 * 4));
    std::vector<float> hf(4096);
    for (auto &v : hf) v = (rand() / (float)RAND_MAX - 0.5f) * 2.0f;
    CHECK(hipMemcpy(fsrc, hf.data(), 4096 * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int g32 = 2048, it32 = 400;       // 4 waves per workgroup, 8 workgroups per CU
    for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(mfma_f32_kernel, dim3(g32), dim3(256), 0, 0, fsrc, sink, it32);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(mfma_f32_kernel, dim3(g32), dim3(256), 0, 0, fsrc, sink, it32);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    const double ms32 = elapsed_ms(e0, e1) / 200;
    const double tf32 = (double)g32 * 4 * it32 * 16.0 * (2.0 * 16 * 16 * 4) / (ms32 * 1e-3) / 1e12;
    // HBM copy
    const size_t bytes = (size_t)1 << 30;
    float4v *ci, *co; CHECK(hipMalloc(&ci, bytes)); CHECK(hipMalloc(&co, bytes));
    CHECK(hipMemset(ci, 1, bytes));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(copy_kernel, dim3(256 * 16), dim3(256), 0, 0, ci, co, bytes / 16);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(copy_kernel, dim3(256 * 16), dim3(256), 0, 0, ci, co, bytes / 16);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    const double tbs = 2.0 * bytes * 10 / (elapsed_ms(e0, e1) * 1e-3) / 1e12;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(read_kernel, dim3(256 * 16), dim3(256), 0, 0, ci, sink, bytes / 16);
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(read_kernel, dim3(256 * 16), dim3(256), 0, 0, ci, sink, bytes / 16);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    const double tbs_rd = 1.0 * bytes * 10 / (elapsed_ms(e0, e1) * 1e-3) / 1e12;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("{\"device\": \"%s\", \"cus\": %d, \"mfma_f16_tflops_sustained\": %.1f, \"mfma_f16_clock_ghz\": %.3f, \"mfma_f16_us_per_launch\": %.1f, "
           "\"mfma_f16_lds_fed_tflops\": %.1f, \"mfma_f16_lds_fed_clock_ghz\": %.3f, "
           "\"mfma_f16_zero_operands_tflops\": %.1f, \"mfma_f16_zero_operands_clock_ghz\": %.3f, "
           "\"mfma_f32_tflops_sustained\": %.1f, \"hbm_copy_tb_s\": %.2f, \"hbm_read_tb_s\": %.2f, "
           "\"lds_fed_tile_variants_tflops\": {\"4x4_16waves\": %.1f, \"8x4_8waves_1wg\": %.1f, \"8x4_2x4waves\": %.1f, \"4x8_2x4waves\": %.1f, \"4x7_8waves_1wg\": %.1f}, "
           "\"note\": \"tools/probes/peak_probe.hip: v_mfma_f32_16x16x32_f16 on random register operands, 8 waves x 2 workgroups per CU, 16 accumulators per wave, "
           "launches of ~100 us back to back for >= 1.5 s (the clock the chip holds under matrix load); lds_fed = operands re-read from LDS at the tap kernel's "
           "ratio; zero_operands = the same loop on zeros (no data-dependent power); float4 copy of 1 GiB (read + write bytes / time) and read-only sweep of 1 GiB\"}\n",
           prop.name, prop.multiProcessorCount, tf_rand, ghz_rand, us_rand, tf_lds, ghz_lds, tf_zero, ghz_zero, tf32, tbs, tbs_rd, t44_16, t84_8, t84_2x4, t48_2x4, t47_8);
    return 0;
}
