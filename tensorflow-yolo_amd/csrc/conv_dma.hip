// Second-generation implicit-GEMM conv for the heavy fp16 layers: same GEMM view, LDS image and
// epilogue as conv.hip, but
//   * 8 waves (512 threads), one workgroup per CU, block tiles 256x256 / 256x128 / 128x256
//     (couts x pixels) -> 32..48 B/clk/CU of L2->LDS traffic instead of 64 for the 128x128 tile;
//   * operands go global -> LDS directly with `buffer_load_dwordx4 ... lds` (LDS-DMA): no staging
//     VGPRs and no ds_write pass (ds_write_b128 tops out at ~79 B/clk/CU and was the bottleneck of
//     the register-staged kernel).  One wave instruction fills 8 LDS rows x 128 B (lane-linear), the
//     XOR swizzle is applied to the per-lane SOURCE chunk, zero padding / M tail come from the
//     buffer range check (verified on gfx950: out-of-range lanes write zeros to LDS);
//   * an S-stage LDS ring with counted `s_waitcnt vmcnt(N)` and raw `s_barrier`, so the DMA of the
//     next tile(s) stays in flight across the barrier while the MFMAs of the current tile run.
// Per K tile: wait(tile kt landed) -> barrier -> issue DMA of tile kt+S-1 into the stage read at
// kt-1 -> ds_read fragments + MFMAs of tile kt.
#include "conv_common.h"

namespace yolo {

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef __attribute__((address_space(3))) void lds_void;

// One LDS-DMA wave instruction: every lane fetches 16 bytes at byte offset `voff` of the buffer and
// the wave's 1 KiB lands lane-linearly at `lds_dst` (wave-uniform).  The builtin only exists in the
// device pass (its instantiation inside a kernel template fails in the host pass).
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char *lds_dst, uint32_t voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)lds_dst, 16, voff, 0, 0, 0);
#else
    (void)rsrc; (void)lds_dst; (void)voff;
#endif
}

// WM x WN = 8 waves; a wave owns TM*16 couts x TP*16 pixels; S = LDS ring depth.
template <int WM, int WN, int TM, int TP, int S>
__global__ void __launch_bounds__(512) conv_igemm_dma_kernel(const ConvParams p) {
    typedef _Float16 T;
    static_assert(WM * WN == 8, "eight waves per workgroup");
    static_assert(S == 2 || S == 3, "ring depth 2 or 3");
    constexpr int NA = WM * TM * 16;        // couts per block
    constexpr int NB = WN * TP * 16;        // pixels per block
    constexpr int JA = NA / 64;             // DMA wave-instructions per wave per tile (weights)
    constexpr int JB = NB / 64;             //                                        (pixels)
    constexpr int NL = JA + JB;
    constexpr int CH = 4 * TM;
    constexpr int TILE_BYTES = (NA + NB) * 128;
    __shared__ __attribute__((aligned(16))) unsigned char smem[S * TILE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int bid = xcd_remap(blockIdx.x, p.n_blocks);
    const int nt = bid % p.n_tiles_n;
    const int mt = bid / p.n_tiles_n;
    const int n0 = nt * NA;
    const int m0 = mt * NB;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wgt), 0, p.wgt_bytes, 0x00020000);

    // ---- DMA geometry ---------------------------------------------------------------------------
    // wave instruction j of wave w fills row group g = 8 j + w (8 rows x 128 B = 1 KiB, lane-linear):
    // lane -> row 8 g + (lane >> 3), PHYSICAL chunk lane & 7, which must hold LOGICAL chunk
    // phys ^ ((row >> 1) & 7) = phys ^ ((4 (w & 1) + (lane >> 4)) & 7)   (same for every j).
    const int lrow = lane >> 3;
    const uint32_t csw = (uint32_t)(((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) << 4);

    uint32_t a_off[JA];
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int r = (j * 8 + wave) * 8 + lrow;                // LDS row of the weight tile
        const int ws = r / (TM * 16), R = r % (TM * 16);
        const int tm = R >> 4, g4 = (R >> 2) & 3, jj = R & 3;
        const int ch = ws * (TM * 16) + g4 * CH + 4 * tm + jj;  // the cout that LDS row holds
        a_off[j] = (uint32_t)(n0 + ch) * p.wrow_bytes + csw;
    }
    uint32_t b_base[JB], b_mask[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int m = m0 + (j * 8 + wave) * 8 + lrow;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = mm / p.HoWo;
        const int rem = mm - n * p.HoWo;
        const int oy = rem / p.Wo;
        const int ox = rem - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        const long long e = (long long)n * p.in_img_stride + ((long long)iy0 * p.W + ix0) * p.in_ld + p.in_coff;
        b_base[j] = (uint32_t)(e * 2) + csw;
        uint32_t mask = 0;
        if (ok) {
            for (int t = 0; t < p.taps; ++t) {
                const int kh = t / p.ksize, kw = t - kh * p.ksize;
                if ((unsigned)(iy0 + kh) < (unsigned)p.H && (unsigned)(ix0 + kw) < (unsigned)p.W) mask |= 1u << t;
            }
        }
        b_mask[j] = mask;
    }

    auto issue_tile = [&](int kt, int stage) {
        unsigned char *base = smem + stage * TILE_BYTES + wave * 1024;
        const uint32_t ka = (uint32_t)kt * 128;
#pragma unroll
        for (int j = 0; j < JA; ++j)
            dma16(rs_w, base + j * 8192, a_off[j] + ka);
        const int tap = kt / p.tiles_per_tap;
        const uint32_t koff = (uint32_t)(kt - tap * p.tiles_per_tap) * 128;
        const int kh = p.ksize == 3 ? (tap * 11) >> 5 : 0;
        const int kw = tap - kh * p.ksize;
        const uint32_t toff = (uint32_t)((kh * p.W + kw) * p.in_ld * 2) + koff;
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const bool ok = (b_mask[j] >> tap) & 1u;
            dma16(rs_in, base + NA * 128 + j * 8192, ok ? b_base[j] + toff : YOLO_INVALID_OFF);
        }
    };

    float4v acc[TM][TP];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = float4v{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int stage) {
        const unsigned char *A = smem + stage * TILE_BYTES + (wm * TM * 16 + fr) * 128;
        const unsigned char *B = smem + stage * TILE_BYTES + NA * 128 + (wn * TP * 16 + fr) * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int so = (((ks * 4 + fq) ^ (fr >> 1)) & 7) << 4;
            uint4v fa[TM], fb[TP];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = *reinterpret_cast<const uint4v *>(A + a * 16 * 128 + so);
#pragma unroll
            for (int b = 0; b < TP; ++b) fb[b] = *reinterpret_cast<const uint4v *>(B + b * 16 * 128 + so);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TP; ++b) acc[a][b] = mma_chunk<T>(fa[a], fb[b], acc[a][b]);
        }
    };

    // ---- main loop: S-stage ring, counted vmcnt, raw barrier ------------------------------------
    const int KT = p.ktiles;
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < KT) issue_tile(s, s);
    int stage = 0;                  // stage holding tile kt
    int fill = S - 1;               // stage that tile kt+S-1 goes to (== the stage read at kt-1)
    for (int kt = 0; kt < KT; ++kt) {
        // tiles issued after tile kt and still allowed in flight: min(S-2, KT-1-kt)
        if (S == 3 && kt + 1 < KT) wait_vmcnt<NL>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();       // tile kt visible to every wave; stage `fill` no longer read
        if (kt + S - 1 < KT) issue_tile(kt + S - 1, fill);
        compute(stage);
        stage = stage + 1 == S ? 0 : stage + 1;
        fill = fill + 1 == S ? 0 : fill + 1;
    }

    conv_epilogue<T, TM, TP>(p, acc, n0 + wm * (TM * 16) + fq * CH, m0 + wn * (TP * 16), fr);
}

struct DmaCfg {
    int na, nb, slots_per_cu;
    float rate;     // relative per-CU throughput while busy (measured ordering, refined by profiling)
};
static const DmaCfg kCfgs[] = {
    {128, 128, 2, 0.55f},   // 0: conv.hip 4-wave register-staged kernel (two workgroups per CU)
    {256, 256, 1, 1.00f},   // 1
    {256, 128, 1, 0.85f},   // 2: 256 couts x 128 pixels
    {128, 256, 1, 0.85f},   // 3: 128 couts x 256 pixels
};

// Pick the block tile that minimises rounds x tile time on 256 CUs (tail quantisation matters:
// e.g. 38x38x512 at batch 32 is 362 tiles of 256x256 = 2 rounds at 71 % but 722 of 256x128 = 3 at 94 %).
int choose_dma_cfg(int M, int cout) {
    const char *force = getenv("YOLO_CONV_TILE");
    if (force && *force) return atoi(force);
    const int cout_pad = (cout + 127) / 128 * 128;
    int best = 0;
    double best_t = 1e300;
    for (int c = 0; c < 4; ++c) {
        const DmaCfg &k = kCfgs[c];
        if (k.na > cout_pad) continue;
        const long long blocks = ((long long)M + k.nb - 1) / k.nb * ((cout + k.na - 1) / k.na);
        const long long rounds = (blocks + 256LL * k.slots_per_cu - 1) / (256LL * k.slots_per_cu);
        const double t = (double)rounds * k.na * k.nb * k.slots_per_cu / k.rate;
        if (t < best_t) { best_t = t; best = c; }
    }
    return best;
}

const char *dma_cfg_name(int cfg) {
    switch (cfg) {
    case 1: return "256x256,S2";
    case 2: return "256x128,S3";
    case 3: return "128x256,S3";
    default: return "";
    }
}

hipError_t launch_conv_dma(const ConvParams &p0, int cfg, hipStream_t s) {
    ConvParams p = p0;
    const DmaCfg &k = kCfgs[cfg];
    p.n_tiles_n = (p.Cout + k.na - 1) / k.na;
    const long long blocks = ((long long)p.M + k.nb - 1) / k.nb * p.n_tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    p.n_blocks = (int)blocks;
    const dim3 grid((unsigned)blocks), block(512);
    switch (cfg) {
    case 1: hipLaunchKernelGGL((conv_igemm_dma_kernel<2, 4, 8, 4, 2>), grid, block, 0, s, p); break;
    case 2: hipLaunchKernelGGL((conv_igemm_dma_kernel<4, 2, 4, 4, 3>), grid, block, 0, s, p); break;
    case 3: hipLaunchKernelGGL((conv_igemm_dma_kernel<2, 4, 4, 4, 3>), grid, block, 0, s, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace yolo
