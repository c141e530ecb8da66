// Second-generation implicit-GEMM conv for the heavy fp16 layers: same GEMM view and epilogue as
// conv.hip, but
//   * 8 waves (512 threads), one workgroup per CU, block tiles 256x256 / 256x128 / 128x256
//     (couts x pixels) -> fewer L2->LDS bytes per FLOP than the 128x128 tile;
//   * operands go global -> LDS directly with `buffer_load_dwordx4 ... lds` (LDS-DMA): no staging
//     VGPRs and no ds_write pass (ds_write_b128 tops out at ~79 B/clk/CU and was the bottleneck of
//     the register-staged kernel).  One wave instruction fills 1 KiB of LDS rows (lane-linear), the
//     XOR swizzle is applied to the per-lane SOURCE chunk, zero padding / M tail come from the
//     buffer range check (verified on gfx950: out-of-range lanes write zeros to LDS);
//   * an S-stage LDS ring with counted `s_waitcnt vmcnt(N)` and raw `s_barrier`, so the DMA of the
//     next tile(s) stays in flight across the barrier while the MFMAs of the current tile run.
//     PMC showed the loop is bound by DMA round-trip latency (~5k cycles under load): throughput =
//     bytes in flight / latency, so the K depth of a stage (BKC chunks: 64 or 32 halfs) and S are
//     chosen to keep as much of the 160 KiB LDS in flight as possible (256x256: 4 stages of K=32).
// Per K tile: wait(tile kt landed) -> barrier -> issue DMA of tile kt+S-1 into the stage read at
// kt-1 -> ds_read fragments + MFMAs of tile kt.
#include "conv_common.h"
#include <cstdio>
#include <vector>

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

// LDS image: rows of BKC 16-byte chunks (128 B or 64 B), logical chunk c of row r stored at physical
// chunk c ^ swz(r).  Chosen so that the ds_read_b128 fragment reads (16 rows x one chunk per 16-lane
// group, hardware lane groups {0-3,12-15,20-27} ...) hit 16 distinct 16-byte slots of the 256-byte
// bank window:   BKC = 8: swz = (r >> 1) & 7;   BKC = 4: swz = {0,2,3,1}[(r >> 2) & 3].
template <int BKC>
__device__ __forceinline__ int lds_swz(int r) {
    if (BKC == 8) return (r >> 1) & 7;
    return (0x78 >> (2 * ((r >> 2) & 3))) & 3;
}

// WM x WN = 8 waves; a wave owns TM*16 couts x TP*16 pixels; S = LDS ring depth; BKC = K chunks per stage;
// OCC = waves per SIMD the register budget must allow (2: one workgroup per CU, 4: two per CU, so that
// one workgroup's epilogue -- ~190 MB of residual reads + output writes per 76x76 layer -- overlaps the
// other's MFMA loop; with one lock-stepped workgroup per CU that traffic was 44 % of the layer time).
// FUSE2: the 1x1 conv behind this one computed by the same workgroups (conv_common.h: conv_epilogue_fused_1x1); the 128 x 256 K32
// tile only (all 128 couts of 256 pixels in one workgroup).
// EPI: which epilogue this instantiation carries -- 0 the generic one (any output map / view), 1 the lean one of conv_common.h
// (conv_epilogue_fast: plain fp16 output maps), 2 the head convs' float32 rows through LDS slabs (conv_epilogue_f32_staged).
// Instantiations of their own, like in conv_tap.hip: with the three behind run-time branches in one kernel seven of the ten tiles
// spilled 12-176 registers (the staged epilogue's row flags alone cost the generic path its last registers).
template <int WM, int WN, int TM, int TP, int S, int BKC, int OCC, bool FUSE2 = false, int EPI = 0>
__global__ void __launch_bounds__(512, OCC) conv_igemm_dma_kernel(const ConvParams p) {
    typedef _Float16 T;
    static_assert(WM * WN == 8, "eight waves per workgroup");
    static_assert(S >= 2 && S <= 4, "ring depth 2..4");
    static_assert(BKC == 8 || BKC == 4, "stage depth 64 or 32 halfs");
    constexpr int ROWB = BKC * 16;          // bytes per LDS row
    constexpr int RPI = 1024 / ROWB;        // rows one DMA wave-instruction fills (8 or 16)
    constexpr int NA = WM * TM * 16;        // couts per block
    constexpr int NB = WN * TP * 16;        // pixels per block
    constexpr int JA_TOT = NA / RPI;        // weight DMA wave-instructions per stage, dealt round-robin to the waves
    constexpr int JA = (JA_TOT + 7) / 8;    // per wave (waves >= JA_TOT carry none when the weight tile is small)
    constexpr int JB = NB / (8 * RPI);      // pixel DMA wave-instructions per wave per stage
    constexpr bool A_ALL = JA_TOT % 8 == 0; // every wave issues the same number of weight instructions
    constexpr int NL = JA + JB;             // DMA instructions per stage of a wave that carries weights
    constexpr int KS = BKC / 4;             // 32-deep MFMA k-steps per stage
    constexpr int CH = 4 * TM;
    constexpr int TILE_BYTES = (NA + NB) * ROWB;
    static_assert(JB >= 1 && (A_ALL || JA_TOT < 8), "unsupported tile for the DMA mapping");
    constexpr int LDS_BYTES = FUSE2 && kFuse2LdsBytes > S * TILE_BYTES ? kFuse2LdsBytes : S * TILE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];

    const int tid = threadIdx.x;
#ifdef YOLO_EXPERIMENT      // block trace (tools/trace_blocks.py); not in the product build
    const unsigned long long t_start = p.trace ? wall_clock64() : 0ull;
    const unsigned long long c_start = p.trace ? (unsigned long long)clock64() : 0ull;
    unsigned long long t_first = 0ull;
#endif
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const bool has_a = A_ALL || wave < JA_TOT;     // wave-uniform

    const int bid = xcd_remap(blockIdx.x, p.n_blocks);
    const int mt = (int)fdiv((uint32_t)bid, p.dtiles_n);
    const int nt = bid - mt * p.n_tiles_n;
    const int n0 = nt * NA;
    const int m0 = mt * NB;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wgt), 0, p.wgt_bytes, 0x00020000);

    // ---- DMA geometry ---------------------------------------------------------------------------
    // wave instruction j of wave w fills row group g = 8 j + w (RPI rows = 1 KiB, lane-linear):
    // lane -> row RPI g + lane / BKC, PHYSICAL chunk lane % BKC, which must hold LOGICAL chunk
    // phys ^ swz(row).  swz(row) only depends on (w & 1, lane) [BKC 8] or lane [BKC 4]: same for all j.
    const int lrow = lane / BKC;
    const uint32_t csw = (uint32_t)(((lane % BKC) ^ lds_swz<BKC>(RPI * (wave & 1) + lrow)) << 4);

    uint32_t a_off[JA];
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int r = (j * 8 + wave) * RPI + lrow;              // LDS row of the weight tile
        const int ws = r / (TM * 16), R = r % (TM * 16);
        const int tm = R >> 4, g4 = (R >> 2) & 3, jj = R & 3;
        const int ch = ws * (TM * 16) + g4 * CH + 4 * tm + jj;  // the cout that LDS row holds
        a_off[j] = (uint32_t)(n0 + ch) * p.wrow_bytes + csw;
    }
    uint32_t b_base[JB], b_mask[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int m = m0 + (j * 8 + wave) * RPI + lrow;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = (int)fdiv((uint32_t)mm, p.dHoWo);
        const int rem = mm - n * p.HoWo;
        const int oy = (int)fdiv((uint32_t)rem, p.dWo);
        const int ox = rem - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        const long long e = (long long)n * p.in_img_stride + ((long long)iy0 * p.W + ix0) * p.in_ld + p.in_coff;
        b_base[j] = (uint32_t)(e * 2) + csw;
        uint32_t mask = 0;
        if (ok) {
            for (int t = 0; t < p.taps; ++t) {
                const int kh = p.ksize == 3 ? (t * 11) >> 5 : 0, kw = t - kh * p.ksize;    // ksize is 1 or 3
                if ((unsigned)(iy0 + kh) < (unsigned)p.H && (unsigned)(ix0 + kw) < (unsigned)p.W) mask |= 1u << t;
            }
        }
        b_mask[j] = mask;
    }

    const int tpt = p.cin_chunks / BKC;     // stages per tap (Cin is a multiple of BKC chunks)
    auto issue_tile = [&](int kt, int stage) {
        unsigned char *base = smem + stage * TILE_BYTES + wave * 1024;
        const uint32_t ka = (uint32_t)kt * ROWB;
        if (has_a) {
#pragma unroll
            for (int j = 0; j < JA; ++j)
                dma16(rs_w, base + j * 8192, a_off[j] + ka);
        }
        const int tap = (int)fdiv((uint32_t)kt, p.dtpt);
        const uint32_t koff = (uint32_t)(kt - tap * tpt) * ROWB;
        const int kh = p.ksize == 3 ? (tap * 11) >> 5 : 0;
        const int kw = tap - kh * p.ksize;
        const uint32_t toff = (uint32_t)((kh * p.W + kw) * p.in_ld * 2) + koff;
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const bool ok = (b_mask[j] >> tap) & 1u;
            dma16(rs_in, base + NA * ROWB + j * 8192, ok ? b_base[j] + toff : YOLO_INVALID_OFF);
        }
    };

    const int fr = lane & 15, fq = lane >> 4;
    float4v acc[TM][TP];
    conv_init_acc_bias<TM, TP>(p, acc, n0 + wm * (TM * 16) + fq * CH);     // the accumulators start from the bias (conv_common.h)
    const int fswz = lds_swz<BKC>(fr);      // fragment rows are 16-aligned + fr
    auto compute = [&](int stage) {
        const unsigned char *A = smem + stage * TILE_BYTES + (wm * TM * 16 + fr) * ROWB;
        const unsigned char *B = smem + stage * TILE_BYTES + NA * ROWB + (wn * TP * 16 + fr) * ROWB;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int so = (((ks * 4 + fq) ^ fswz) & (BKC - 1)) << 4;
            uint4v fa[TM], fb[TP];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = *reinterpret_cast<const uint4v *>(A + a * 16 * ROWB + so);
#pragma unroll
            for (int b = 0; b < TP; ++b) fb[b] = *reinterpret_cast<const uint4v *>(B + b * 16 * ROWB + so);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TP; ++b) acc[a][b] = mma_chunk<T>(fa[a], fb[b], acc[a][b]);
        }
    };

    // ---- main loop: S-stage ring, counted vmcnt, raw barrier ------------------------------------
    const int KT = p.taps * tpt;
#ifdef YOLO_EXPERIMENT
    const unsigned long long t_setup = p.trace ? wall_clock64() : 0ull;
#endif
#pragma unroll
    for (int s = 0; s < S - 1; ++s)
        if (s < KT) issue_tile(s, s);
    int stage = 0;                  // stage holding tile kt
    int fill = S - 1;               // stage that tile kt+S-1 goes to (== the stage read at kt-1)
    for (int kt = 0; kt < KT; ++kt) {
        // tiles issued after tile kt that may stay in flight: min(S-2, KT-1-kt)
        const int after = KT - 1 - kt;
        if (has_a) {
            if (S >= 4 && after >= 2) wait_vmcnt<2 * NL>();
            else if (S >= 3 && after >= 1) wait_vmcnt<NL>();
            else wait_vmcnt<0>();
        } else {
            if (S >= 4 && after >= 2) wait_vmcnt<2 * JB>();
            else if (S >= 3 && after >= 1) wait_vmcnt<JB>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_sched_barrier(0);  // no ds_read / MFMA of the previous tile moves below the barrier (see conv_tap.hip)
        __builtin_amdgcn_s_barrier();       // tile kt visible to every wave; stage `fill` no longer read
#ifdef YOLO_EXPERIMENT      // ablation flags (tools/ablate.py: results intentionally wrong); not in the product build
        if (p.trace && kt == 0) t_first = wall_clock64();
        if (kt + S - 1 < KT && !(p.dbg & 1)) issue_tile(kt + S - 1, fill);
        if (!(p.dbg & 2)) compute(stage);
#else
        if (kt + S - 1 < KT) issue_tile(kt + S - 1, fill);
        compute(stage);
#endif
        stage = stage + 1 == S ? 0 : stage + 1;
        fill = fill + 1 == S ? 0 : fill + 1;
    }

#ifdef YOLO_EXPERIMENT
    if (p.dbg & 4) return;             // experiment: no epilogue
    const unsigned long long t_loop = p.trace ? wall_clock64() : 0ull;
#endif
    if constexpr (FUSE2) {
        static_assert(!FUSE2 || (WM == 2 && WN == 4 && TM == 4 && TP == 4 && 2 * LDS_BYTES <= 163840), "back-to-back 1x1: 128 x 256 tile, two per CU");
        conv_epilogue_fused_1x1<0, false>(p, acc, m0, wm, wn, wave, lane, smem);      // (the stride-2 conv into a stage: no residual)
    } else
    if constexpr (EPI == 2) {      // head conv: coalesced float32 rows via LDS
        static_assert(8 * 16 * kStagePitch(TM) * 4 + NB * kStageFlagAnchors * 4 <= S * TILE_BYTES, "staging slabs + row flags must fit in the ring");
        __syncthreads();            // every wave is done reading the ring
        conv_epilogue_f32_staged<TM, TP, 0, true, NB>(p, acc, n0 + wm * (TM * 16), m0 + wn * (TP * 16), lane,
                                         reinterpret_cast<float *>(smem) + wave * 16 * kStagePitch(TM),
                                         reinterpret_cast<float *>(smem) + 8 * 16 * kStagePitch(TM), wn * (TP * 16));
    } else if constexpr (EPI == 1) {    // (plain fp16 output map, aligned views below 2 GiB)
        conv_epilogue_fast<TM, TP, 0>(p, acc, n0 + wm * (TM * 16) + fq * CH, m0 + wn * (TP * 16), fr);
    } else {
        conv_epilogue<T, TM, TP, 0, true>(p, acc, n0 + wm * (TM * 16) + fq * CH, m0 + wn * (TP * 16), fr);
    }
#ifdef YOLO_EXPERIMENT
    if (p.trace && tid == 0) {          // YOLO_CONV_TRACE: phase timestamps (100 MHz) + placement of wave 0 of every block
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *r = p.trace + (size_t)blockIdx.x * 8;
        r[0] = t_start; r[1] = t_setup; r[2] = t_loop; r[3] = wall_clock64();
        r[4] = __builtin_amdgcn_s_getreg(0xF804);      // HW_ID
        r[5] = __builtin_amdgcn_s_getreg(0xF814);      // XCC_ID
        r[6] = t_first;                                // first K tile landed (prologue DMA latency)
        r[7] = (unsigned long long)clock64() - c_start;
    }
#endif
}

struct DmaCfg {
    int na, nb, slots_per_cu;
    float rate;     // relative per-CU throughput while busy (measured ordering, refined by profiling)
    const char *name;
    int bkc;        // K chunks per stage: Cin must be a multiple of it
};
static const DmaCfg kCfgs[] = {
    {128, 128, 2, 0.55f, "", 8},                   // 0: conv.hip 4-wave register-staged kernel (two workgroups per CU)
    {256, 256, 1, 0.80f, "256x256,K64,S2", 8},     // 1
    {256, 128, 1, 0.85f, "256x128,K64,S3", 8},     // 2: 256 couts x 128 pixels
    {128, 256, 1, 0.85f, "128x256,K64,S3", 8},     // 3: 128 couts x 256 pixels
    {256, 256, 1, 1.00f, "256x256,K32,S4", 4},     // 4: 96 KiB in flight instead of 64
    {256, 128, 2, 1.30f, "256x128,K32,S3,x2", 4},  // 5: 72 KiB LDS, <=128 VGPRs: two workgroups per CU
    {128, 256, 2, 1.30f, "128x256,K32,S3,x2", 4},  // 6
    {64, 512, 2, 1.00f, "64x512,K32,S2,x2", 4},    // 7: narrow early layers (Cout <= 64), bandwidth-bound
    {128, 256, 2, 1.00f, "128x256,tap9,x2", 4},        // 8: conv_tap.hip, 3x3/1 only: input patch loaded once for the 9 taps
    {256, 256, 1, 1.00f, "256x256,tap9", 4},           // 9
    {128, 192, 2, 1.00f, "128x192,tap9,x2", 4},        // 10: smaller position tiles for small feature maps
    {128, 128, 2, 1.00f, "128x128,tap9,x2", 4},        // 11
    {128, 256, 2, 1.00f, "128x256,tap9,2d,x2", 4},     // 12: 2-D 16x16 tiles for maps wider than 78
    {64, 256, 2, 1.00f, "64x256,tap9,2d,x2", 4},       // 13: ... and Cout <= 64
    {128, 128, 3, 1.00f, "128x128,K32,S3,x3", 4},      // 14: conv_dma again: 48 KiB LDS, <= 80 VGPRs: three workgroups per CU (short-K 1x1 layers)
    {256, 224, 1, 1.00f, "256x224,tap9", 4},           // 15: conv_tap.hip variant 6 (see there)
    {128, 128, 3, 1.00f, "128x128,tap9,2d,x3", 4},     // 16: conv_tap.hip variant 7: 8 x 16 2-D tile, three workgroups per CU
    {32, 256, 2, 1.00f, "32x256,tap9,2d,x2", 4},       // 17: conv_tap.hip variant 8: 32 couts x (16 x 16)
    {128, 384, 1, 1.00f, "128x384,tap9,img", 4},       // 18: conv_tap.hip variant 9: one whole image (19 x 19) per tile
    {128, 192, 1, 1.00f, "128x192,K64,S4", 8},         // 19: conv_dma again: the whole LDS as a four-stage ring (120 KiB in flight), for one-round 1x1 layers on small maps
    {128, 256, 2, 1.00f, "128x256,tap9,s2,x2", 4},     // 20: conv_tap.hip variant 10: 3x3 / stride 2 with tap reuse over the input's parity planes
    {128, 384, 1, 1.00f, "128x384,tap9,s2,img", 4},    // 21: conv_tap.hip variant 11: ... one whole (19 x 19) output image per tile
    {128, 192, 2, 1.00f, "128x192,tap9,img,x2", 4},    // 22: conv_tap.hip variant 12: one whole 12 x 12 / 13 x 13 image per tile (stride 1)
    {128, 256, 2, 1.00f, "128x256,tap9,s2,wide,x2", 4},   // 23: conv_tap.hip variant 13: tile 20 for output maps up to 158 wide; hosts the back-to-back 1x1
};
static const int kNumCfgs = 24;
static const int kFirstTapCfg = 8, kLastTapCfg = 13;
static inline bool is_tap_cfg(int cfg) { return (cfg >= kFirstTapCfg && cfg <= kLastTapCfg) || (cfg >= 15 && cfg <= 18) || (cfg >= 20 && cfg <= 23); }
static inline int tap_variant(int cfg) { return cfg >= 20 ? cfg - 10 : cfg >= 15 ? cfg - 9 : cfg - kFirstTapCfg; }     // conv_tap.hip variant of a tap cfg
bool dma_cfg_is_tap(int cfg) { return is_tap_cfg(cfg); }
bool dma_cfg_f32_ok(int cfg) { return is_tap_cfg(cfg) && conv_tap_f32_ok(tap_variant(cfg)); }

// Pick the block tile that minimises rounds x tile time on 256 CUs (tail quantisation matters:
// e.g. 38x38x512 at batch 32 is 362 tiles of 256x256 = 2 rounds at 71 % but 722 of 256x128 = 3 at 94 %).
bool dma_cfg_valid(int cfg, int cout, int cin_chunks, bool v1_ok, int ksize, int stride, int W) {
    if (cfg == 0) return v1_ok;
    if (cfg < 0 || cfg >= kNumCfgs) return false;
    const DmaCfg &k = kCfgs[cfg];
    if (cin_chunks % k.bkc) return false;
    if (is_tap_cfg(cfg) && (ksize != 3 || stride != (conv_tap_stride2(tap_variant(cfg)) ? 2 : 1) || !conv_tap_fits(tap_variant(cfg), W))) return false;
    if (k.na == 32) return cout <= 32 && cout > 16;
    if (k.na == 64) return cout <= 64 && (!is_tap_cfg(cfg) || cout > 32);
    return k.na <= (cout + 127) / 128 * 128 && cout > 64;
}

// Default tile per layer: a cost model calibrated on MI355X (profiles/r01_ablation.md).  A launch runs
// ceil(workgroups / resident slots) rounds; a round costs `a` microseconds per 64-deep K tile (MFMA +
// DMA of the tile) plus a fixed `f` (setup, prologue DMA latency, epilogue traffic; ~30 % less without a
// residual to read).  The constants reproduce the measured layer times of YOLOv3-608/416 and YOLOv2-416
// within ~10 %, e.g. 13x13x512->1024 at batch 16: 124 us predicted on 256x256 (123 measured, 44
// workgroups on 256 CUs) vs 71 us on 256x128,K64 (75 measured).  yolo_net_autotune replaces the model by
// on-device timing when asked.
// Two-per-CU tiles have three regimes: more workgroups than the 512 slots (back-filled rounds, `a_shared`), between
// 257 and 512 (one round, most CUs shared and in lockstep: `a_mid`), at most 256 (a workgroup has its CU alone).
struct TileCost { float a_shared, a_mid, a_alone, f; };   // us per K64 tile, fixed us per round
static const TileCost kCost[] = {
    {1.14f, 1.14f, 0.82f, 9.0f},        // 0: 4-wave 128x128, two per CU
    {1.45f, 1.45f, 1.45f, 20.0f},       // 1: 256x256 K64 S2
    {0.82f, 0.82f, 0.82f, 12.0f},       // 2: 256x128 K64 S3
    {0.82f, 0.82f, 0.82f, 12.0f},       // 3: 128x256 K64 S3
    {1.47f, 1.47f, 1.47f, 20.0f},       // 4: 256x256 K32 S4
    {1.75f, 1.75f, 1.30f, 8.0f},        // 5: 256x128 K32 S3, two per CU
    {1.75f, 1.75f, 1.30f, 8.0f},        // 6: 128x256 K32 S3, two per CU
    {0.0f, 0.0f, 0.0f, 0.0f},           // 7: 64x512 (bandwidth-bound narrow layers: chosen by rule)
    {1.20f, 1.45f, 0.76f, 7.9f},        // 8: 128x256 tap reuse, two per CU   (fitted on the sweep of tools/gpu_tile_sweep.sh:
    {1.20f, 1.20f, 1.20f, 20.0f},       // 9: 256x256 tap reuse                v3-608-b32, v3-416-b32, v2-416-b16, within ~8 %)
    {0.94f, 1.32f, 0.68f, 8.5f},        // 10: 128x192 tap reuse, two per CU
    {0.76f, 0.90f, 0.63f, 6.9f},        // 11: 128x128 tap reuse, two per CU
    {0.90f, 1.10f, 0.60f, 17.0f},       // 12: 128x256 tap reuse, 2-D tiles (152x152 64->128: 175 us, 76x76: 117, 38x38: 138)
    {1.00f, 1.00f, 0.60f, 5.8f},        // 13: 64x256 tap reuse, 2-D tiles (chosen by rule below)
    {1.10f, 1.50f, 0.70f, 8.0f},        // 14: 128x128 K32 S3, three per CU: 1x1 layers only (short K, memory / latency bound)
    {1.05f, 1.05f, 1.05f, 18.0f},       // 15: 256x224 tap reuse (7/8 of the 256x256 tile's loop)
    {0.0f, 0.0f, 0.0f, 0.0f},           // 16: 128x128 2-D tap reuse, three per CU (chosen by rule)
    {0.0f, 0.0f, 0.0f, 0.0f},           // 17: 32x256 2-D tap reuse (chosen by rule)
    {0.90f, 0.90f, 0.90f, 18.0f},       // 18: 128x384 image-aligned tap reuse (6/7 of the 256x224 tile's loop)
    {0.60f, 0.60f, 0.60f, 12.0f},       // 19: 128x192 K64 S4
    {1.20f, 1.45f, 0.76f, 7.9f},        // 20: 128x256 stride-2 tap reuse, two per CU (as tile 8)
    {0.90f, 0.90f, 0.90f, 18.0f},       // 21: 128x384 image-aligned stride-2 tap reuse (as tile 18)
    {0.94f, 1.32f, 0.68f, 8.5f},        // 22: 128x192 image-aligned tap reuse, two per CU (as tile 10)
    {0.0f, 0.0f, 0.0f, 0.0f},           // 23: wide stride-2 tap reuse (chosen by rule)
};

int choose_dma_cfg(int M, int cout, int cin_chunks, int taps, int has_res, bool v1_ok, int stride, int W, bool tap_only) {
    const int ksize = taps == 9 ? 3 : 1;
#ifdef YOLO_EXPERIMENT      // tools/gpu_tile_sweep.sh: force one tile id on every layer that accepts it
    const char *force = getenv("YOLO_CONV_TILE");
    if (force && *force && dma_cfg_valid(atoi(force), cout, cin_chunks, v1_ok, ksize, stride, W) && (!tap_only || atoi(force) == 0 || is_tap_cfg(atoi(force))))
        return atoi(force);
#endif
    const int fallback = v1_ok ? 0 : -1;
    if (cout <= 32 && M >= 8192 && dma_cfg_valid(17, cout, cin_chunks, v1_ok, ksize, stride, W)) return 17;     // (tiny-YOLOv2 16 -> 32 at 208 x 208)
    if (cout <= 64) {   // narrow, bandwidth-bound layers: 3x3/1 with tap reuse (304x304 32->64: 239 us vs 273 on the 64x512 tile)
        if (M >= 8192 && dma_cfg_valid(13, cout, cin_chunks, v1_ok, ksize, stride, W)) return 13;
        return !tap_only && dma_cfg_valid(7, cout, cin_chunks, v1_ok, ksize, stride, W) && M >= 8192 ? 7 : fallback;
    }
    // wide maps with a short K (152x152 64 -> 128 at batch 32, round 3, after the bias moved into the accumulators and the 2-D
    // tile lost its 36 spilled registers: 155 us on the 128 x 256 2-D tap tile, 161 on the 128 x 128 2-D tile at three
    // workgroups per CU, 175 on the padded-linear 128 x 128 tile, 181 on the per-tap LDS-DMA tile the model picks)
    if (!tap_only && taps == 9 && stride == 1 && W > 110 && cout <= 128 && cin_chunks <= 8 && M >= 262144 &&
        dma_cfg_valid(12, cout, cin_chunks, v1_ok, ksize, stride, W))
        return 12;
    // ... and at a quarter of that batch the 8 x 16 tile at three workgroups per CU (152x152 64->128 at batch 8: 40 us vs 47.5 on
    // the per-tap LDS-DMA tile the model picks; at batch 32 it is the slower of the two 2-D tiles, 161 vs 155)
    if (!tap_only && taps == 9 && stride == 1 && W > 110 && cout <= 128 && cin_chunks <= 8 && M >= 65536 && M < 262144 &&
        dma_cfg_valid(16, cout, cin_chunks, v1_ok, ksize, stride, W))
        return 16;
    // float32, MFMA-bound (fp32 matrix peak is 1/16 of fp16's): long K on a 13x13 map over more than one round of workgroups.  There
    // the padded-linear grid's (14/13)^2 = 16 % of computed-and-dropped positions are the whole difference: tiny-YOLOv2 b64
    // 1024->1024 2.09 ms on the tap tile (97 TFLOP/s) vs 1.81 ms on the 4-wave kernel (113 TFLOP/s = 72 % of the fp32 MFMA peak),
    // 512->1024 1.06 vs 0.91 ms; 256->512 (one round, K 2304) and every 26x26 / 52x52 layer stay faster on the tap tile.
    if (tap_only && v1_ok && taps == 9 && stride == 1 && W <= 14 && taps * cin_chunks * 4 >= 4608 &&
        ((long long)M * (W + 1) * (W + 1) / ((long long)W * W) + 127) / 128 * ((cout + 127) / 128) > 512)
        return 0;
    // (tile 23, the parity-plane tap tile with the back-to-back 1x1 for the stride-2 conv 64 -> 128 into a wide stage, is NOT chosen by
    // rule: at 304 -> 152, batch 32, it measures 211 us against 208 for LDS-DMA tile 6 with the same fusion -- two channel slices are
    // too short a K loop for tap reuse to matter, the launch is the 660 MB it moves)
    const double k64 = taps * cin_chunks / 8.0;         // 64-deep K tiles
    int best = fallback;
    double best_t = 1e300;
    for (int c = 0; c < kNumCfgs; ++c) {
        // float32 nets: only conv_tap.hip has a float32 instantiation, and it beats the 4-wave kernel by ~10 % on every 3x3/1
        // layer measured (tiny-YOLOv2 b64: 1024->1024 at 13x13 2.26 -> 2.06 ms), so the 4-wave kernel is only the fallback
        if (tap_only && !dma_cfg_f32_ok(c)) continue;
        if (c == 12 && W <= 110) continue;      // the padded-linear tiles fit and measured faster (104x104: 70 vs 84 us)
        // the three-per-CU tile: 1x1 layers; measured 10-20 % slower than the larger tiles on every 3x3 layer that fills the chip, but
        // the best tile for a stride-2 layer of 129-256 workgroups (19x19 512->1024 at batch 8: 57 us vs 72 on the 4-wave kernel)
        // (with <= 128 workgroups the 4-wave kernel's split-K wins: 25 vs 49 us at batch 1)
        const long long wg128 = (long long)((M + 127) / 128) * ((cout + 127) / 128);
        if (c == 14 && taps != 1 && !(stride == 2 && wg128 > 128 && wg128 <= 256)) continue;
        if (c == 19 && taps != 1) continue;     // (measured on 1x1 layers only)
        if (c == 7 || c == 13 || c == 16 || c == 17 || c == 23 || !dma_cfg_valid(c, cout, cin_chunks, v1_ok, ksize, stride, W)) continue;
        const DmaCfg &k = kCfgs[c];
        // tap-reuse tiles walk the padded position grid: (H+1)(W+1) positions per image (square maps assumed here)
        long long Meff = M;
        const int Wq = is_tap_cfg(c) && conv_tap_stride2(tap_variant(c)) ? W / 2 : W;      // width of the map the tap tiles walk
        if (is_tap_cfg(c) && conv_tap_stride2(tap_variant(c))) {
            if (cin_chunks <= 8) continue;      // (Cin <= 64, two slices: the LDS-DMA tile that also computes the 1x1 behind it is the better launch)
            Meff = (long long)M * (Wq + 1) * (Wq + 1) / ((long long)Wq * Wq);
        } else if (is_tap_cfg(c)) {
            if (conv_tap_is2d(tap_variant(c))) { const long long th = k.nb / 16, tx = (W + 15) / 16, ty = (W + th - 1) / th; Meff = (long long)M * tx * ty * k.nb / ((long long)W * W); }
            else Meff = (long long)M * (W + 1) * (W + 1) / ((long long)W * W);
        }
        long long blocks = (Meff + k.nb - 1) / k.nb * ((cout + k.na - 1) / k.na);
        if (is_tap_cfg(c) && conv_tap_image_aligned(tap_variant(c))) blocks = (long long)(M / (Wq * Wq)) * ((cout + k.na - 1) / k.na);    // a tile per image
        // (a stride-2 launch of a handful of tiles belongs to the 4-wave kernel, which splits K: 38 -> 19 at batch 1 25 us there, 58 us here)
        if (is_tap_cfg(c) && conv_tap_stride2(tap_variant(c)) && blocks <= 128) continue;
        const long long slots = 256LL * k.slots_per_cu;     // resident workgroups on the chip
        // one workgroup per CU: whole rounds; two per CU: the dispatcher back-fills, the tail costs ~half a round
        double rounds, a;
        if (k.slots_per_cu == 1) {
            rounds = (double)((blocks + slots - 1) / slots);
            a = kCost[c].a_shared;
        } else if (blocks > slots) {
            rounds = (double)blocks / slots + 0.5;
            a = kCost[c].a_shared;
        } else {
            rounds = 1.0;
            a = blocks <= 256 ? kCost[c].a_alone : kCost[c].a_mid;
        }
        const double t = rounds * (k64 * a + kCost[c].f * (has_res ? 1.0 : 0.7));
        if (t < best_t) { best_t = t; best = c; }
    }
    return best;
}

int dma_num_cfgs() { return kNumCfgs; }
int dma_cfg_na(int cfg) { return cfg > 0 && cfg < kNumCfgs ? kCfgs[cfg].na : 128; }
int dma_cfg_nb(int cfg) { return cfg > 0 && cfg < kNumCfgs ? kCfgs[cfg].nb : 128; }
bool dma_cfg_splitk_ok(int cfg) { return is_tap_cfg(cfg) && conv_tap_splitk_ok(tap_variant(cfg)); }
int dma_cfg_bkc(int cfg) { return cfg > 0 && cfg < kNumCfgs ? kCfgs[cfg].bkc : 8; }
const char *dma_cfg_name(int cfg) { return cfg > 0 && cfg < kNumCfgs ? kCfgs[cfg].name : ""; }

// tile id, then the template arguments WM, WN, TM, TP, S, BKC, OCC (", "-separated: the stringified list equals the demangled symbol)
#define YOLO_DMA_VARIANTS(X) \
    X(1, 2, 4, 8, 4, 2, 8, 2) \
    X(2, 4, 2, 4, 4, 3, 8, 2) \
    X(3, 2, 4, 4, 4, 3, 8, 2) \
    X(4, 2, 4, 8, 4, 4, 4, 2) \
    X(5, 4, 2, 4, 4, 3, 4, 4) \
    X(6, 2, 4, 4, 4, 3, 4, 4) \
    X(7, 1, 8, 4, 4, 2, 4, 4) \
    X(14, 2, 4, 4, 2, 3, 4, 6) \
    X(19, 2, 4, 4, 3, 4, 8, 2)

// the epilogue instantiation a launch runs (see the kernel's EPI): 2 = float32 head rows, 1 = lean fp16, 0 = generic
static inline int dma_epilogue_kind(const ConvParams &p) {
    if (p.out_f32 && !p.vec_out && p.outmode == OUT_NORMAL && !p.has_res) return 2;
    static const bool no_fast_epi = getenv("YOLO_NO_FAST_EPI") != nullptr;        // A/B switch, read once (same results either way)
    return !no_fast_epi && conv_fast_epilogue_ok(p) ? 1 : 0;
}

static hipError_t launch_dma_tile(const ConvParams &p, int cfg, hipStream_t s) {
    const dim3 grid((unsigned)p.n_blocks), block(512);
    if (p.fuse2) {          // back-to-back 1x1: the 128 x 256 K32 tile
        if (cfg != 6 || !conv_fast_epilogue_ok(p) || p.has_res || p.n_tiles_n != 1 || p.Cout != 128 || !p.w2 || !p.b2 || !p.out2 || !p.out2_bytes) return hipErrorInvalidValue;
        hipLaunchKernelGGL((conv_igemm_dma_kernel<2, 4, 4, 4, 3, 4, 4, true>), grid, block, 0, s, p);
        return hipGetLastError();
    }
    switch (cfg) {
#define X(id, ...) case id: \
        if (dma_epilogue_kind(p) == 2) hipLaunchKernelGGL((conv_igemm_dma_kernel<__VA_ARGS__, false, 2>), grid, block, 0, s, p); \
        else if (dma_epilogue_kind(p) == 1) hipLaunchKernelGGL((conv_igemm_dma_kernel<__VA_ARGS__, false, 1>), grid, block, 0, s, p); \
        else hipLaunchKernelGGL((conv_igemm_dma_kernel<__VA_ARGS__, false, 0>), grid, block, 0, s, p); \
        break;
        YOLO_DMA_VARIANTS(X)
#undef X
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

#ifdef YOLO_EXPERIMENT
// experiment: YOLO_CONV_TRACE=<file> appends, for every LDS-DMA / tap-reuse launch, a header {blocks, M, Cout, cin_chunks, H, W, cfg,
// ksize * 10 + stride} and 8 uint64 per block (see the kernels); synchronous, for tools/trace_blocks.py only
static hipError_t launch_traced(ConvParams p, int cfg, hipStream_t s) {
    const char *tf = getenv("YOLO_CONV_TRACE");
    unsigned long long *dev = nullptr;
    const size_t bytes = (size_t)p.n_blocks * 64;
    if (hipMalloc((void **)&dev, bytes) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipMemset(dev, 0, bytes);
    p.trace = dev;
    hipError_t e = is_tap_cfg(cfg) ? launch_conv_tap(p, tap_variant(cfg), s) : launch_dma_tile(p, cfg, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    std::vector<unsigned long long> host((size_t)p.n_blocks * 8);
    if (e == hipSuccess) e = hipMemcpy(host.data(), dev, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    if (e == hipSuccess) {
        if (FILE *fp = fopen(tf, "ab")) {
            const unsigned long long hdr[8] = {(unsigned long long)p.n_blocks, (unsigned long long)p.M, (unsigned long long)p.Cout,
                                               (unsigned long long)p.cin_chunks, (unsigned long long)p.H, (unsigned long long)p.W,
                                               (unsigned long long)cfg, (unsigned long long)(p.ksize * 10 + p.stride)};
            fwrite(hdr, 8, 8, fp);
            fwrite(host.data(), 8, host.size(), fp);
            fclose(fp);
        }
    }
    return e;
}
#endif

hipError_t launch_conv_dma(const ConvParams &p0, int cfg, hipStream_t s) {
    if (cfg <= 0 || cfg >= kNumCfgs) return hipErrorInvalidValue;
    ConvParams p = p0;
    const DmaCfg &k = kCfgs[cfg];
#ifdef YOLO_EXPERIMENT
    { const char *d = getenv("YOLO_CONV_DBG"); p.dbg = d ? atoi(d) : 0; }
#endif
    p.n_tiles_n = (p.Cout + k.na - 1) / k.na;
    const long long blocks = ((long long)p.M + k.nb - 1) / k.nb * p.n_tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    p.n_blocks = (int)blocks;
    conv_set_divisors(p, p.cin_chunks / k.bkc);
    if (is_tap_cfg(cfg)) {          // padded-linear position grid: one shared pad column per row, one pad row per image
        long long mq;
        if (conv_tap_is2d(tap_variant(cfg))) {     // TH x 16 tiles (TH = positions per block / 16): qW = tiles per tile row, qHW = tiles per image
            const int th = k.nb / 16;
            p.t2_shift = th == 16 ? 8 : 7;
            if (th != 16 && th != 8) return hipErrorInvalidValue;
            p.qW = (p.W + 15) / 16;
            p.qHW = p.qW * ((p.H + th - 1) / th);
            mq = (long long)(p.M / p.HoWo) * p.qHW * k.nb;
        } else {
            p.qW = p.Wo + 1;          // (stride 1: Ho = H, Wo = W; the stride-2 tiles walk the OUTPUT map)
            p.qHW = (p.Ho + 1) * (p.Wo + 1);
            mq = (long long)(p.M / p.HoWo) * p.qHW;
        }
        const bool img = conv_tap_image_aligned(tap_variant(cfg));
        p.q_stride = img ? p.qHW : k.nb;
        const long long qblocks = img ? (long long)(p.M / p.HoWo) * p.n_tiles_n : (mq + k.nb - 1) / k.nb * p.n_tiles_n;
        if (qblocks <= 0 || qblocks > 0x7fffffffLL) return hipErrorInvalidValue;
        p.Mq = (int)mq;
        p.n_blocks = (int)qblocks;
        conv_set_divisors(p, p.cin_chunks / k.bkc);
#ifdef YOLO_EXPERIMENT
        if (getenv("YOLO_CONV_TRACE")) return launch_traced(p, cfg, s);
#endif
        p.stream = 1;       // the persistent form of the tap kernel where it exists and applies (conv_tap.hip: conv_tap_stream_ok)
#ifdef YOLO_EXPERIMENT
        if (getenv("YOLO_NO_TAP_STREAM")) p.stream = 0;
#endif
        return launch_conv_tap(p, tap_variant(cfg), s);
    }
#ifdef YOLO_EXPERIMENT
    if (getenv("YOLO_CONV_TRACE")) return launch_traced(p, cfg, s);
#endif
    return launch_dma_tile(p, cfg, s);
}

std::string dma_cfg_symbol_for(int cfg, bool f32, const ConvParams &p) {
    if (p.fuse2) return cfg == 6 ? "void yolo::conv_igemm_dma_kernel<2, 4, 4, 4, 3, 4, 4, true, 0>(yolo::ConvParams)"
                       : cfg == 23 ? "void yolo::conv3x3_tap_kernel<false, 2, 4, 4, 4, 26, 4, 4, false, true, true>(yolo::ConvParams)"
                                   : "void yolo::conv3x3_tap_kernel<false, 2, 4, 4, 4, 27, 4, 2, false, true, true>(yolo::ConvParams)";
    if (is_tap_cfg(cfg) && conv_tap_stream_ok(p, tap_variant(cfg))) return conv_tap_stream_symbol(tap_variant(cfg));
    if (!is_tap_cfg(cfg)) {     // the LDS-DMA kernel: last template argument = the epilogue kind of this launch
        std::string sym = dma_cfg_symbol(cfg, f32, false);
        const size_t at = sym.rfind(", false, 0>(");
        if (at != std::string::npos) sym[at + 9] = (char)('0' + dma_epilogue_kind(p));
        return sym;
    }
    std::string sym = dma_cfg_symbol(cfg, f32, !f32 && conv_fast_epilogue_ok(p) && cfg != 20 && cfg != 23);
    if (is_tap_cfg(cfg) && p.outmode == OUT_POOL2) {        // the fused-pool instantiation: template argument MODE 3 instead of 2
        const size_t at = sym.rfind(", 2, false, false, false>(");
        if (at != std::string::npos) sym.replace(at, 26, ", 3, false, false, false>(");
    }
    return sym;
}

// the name rocprofv3's kernel trace prints for the kernel a tile id runs (yolo_kernel_info.symbol)
const char *dma_cfg_symbol(int cfg, bool f32, bool fast) {
    if (is_tap_cfg(cfg)) return conv_tap_symbol(tap_variant(cfg), f32, fast);
    switch (cfg) {
#define X(id, ...) case id: return fast ? "void yolo::conv_igemm_dma_kernel<" #__VA_ARGS__ ", false, 1>(yolo::ConvParams)" \
                                          : "void yolo::conv_igemm_dma_kernel<" #__VA_ARGS__ ", false, 0>(yolo::ConvParams)";
        YOLO_DMA_VARIANTS(X)
#undef X
    default: return "";
    }
}

}  // namespace yolo
