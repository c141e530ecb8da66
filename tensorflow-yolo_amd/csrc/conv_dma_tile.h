// The LDS-DMA implicit-GEMM tile of conv_dma.hip as a device function (one workgroup, one tile), shared by its kernel there and by the chained
// launch of conv_chain.hip.  See conv_dma.hip for the design.
#pragma once
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
// AUX: cache policy (0 plain; 16 = sc1, conv_chain.hip).
template <int AUX = 0>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char *lds_dst, uint32_t voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void *)lds_dst, 16, voff, 0, 0, AUX);
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
// (the kernel = one call of conv_igemm_dma_tile, as in conv_tap.hip: the chained launch of conv_chain.hip runs the same code with COH -- the
// pixel operand and the residual read with sc1 loads)
template <int WM, int WN, int TM, int TP, int S, int BKC, bool FUSE2>
constexpr int conv_igemm_dma_lds_bytes() {
    constexpr int ring = S * (WM * TM * 16 + WN * TP * 16) * BKC * 16;
    return FUSE2 && kFuse2LdsBytes > ring ? kFuse2LdsBytes : ring;
}
template <int WM, int WN, int TM, int TP, int S, int BKC, int OCC, bool FUSE2 = false, int EPI = 0, bool COH = false, bool SC1 = COH>
__device__ __forceinline__ void conv_igemm_dma_tile(const ConvParams p, unsigned char *const lds, const int bid, const int raw_x) {
    typedef _Float16 T;
    constexpr int LDAUX = SC1 ? 16 : 0;
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
    constexpr int LDS_BYTES = conv_igemm_dma_lds_bytes<WM, WN, TM, TP, S, BKC, FUSE2>();
    static_assert(LDS_BYTES >= S * TILE_BYTES, "LDS size of the tile");
    // (its own LDS unless the chained launch lends it: with the array passed in by every caller, the head-conv instantiations -- at 127 of 128
    // registers -- spilled 5-7)
    __shared__ __attribute__((aligned(16))) unsigned char own[COH ? 16 : LDS_BYTES];
    unsigned char *const smem = COH ? lds : own;

    int tid_ = threadIdx.x;
    // (the chained launch calls this in a loop: without the empty statement everything derived from the thread index is loop-invariant, gets
    // hoisted and stays live across the whole iteration -- +24 scalar and ~9 vector registers, 6 spilled)
    if constexpr (COH) asm volatile("" : "+v"(tid_));
    const int tid = tid_;
#ifdef YOLO_EXPERIMENT      // block trace (tools/trace_blocks.py); not in the product build
    const unsigned long long t_start = p.trace ? wall_clock64() : 0ull;
    const unsigned long long c_start = p.trace ? (unsigned long long)clock64() : 0ull;
    unsigned long long t_first = 0ull;
#endif
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const bool has_a = A_ALL || wave < JA_TOT;     // wave-uniform

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
            dma16<LDAUX>(rs_in, base + NA * ROWB + j * 8192, ok ? b_base[j] + toff : YOLO_INVALID_OFF);
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
        conv_epilogue_fast<TM, TP, 0, false, LDAUX>(p, acc, n0 + wm * (TM * 16) + fq * CH, m0 + wn * (TP * 16), fr);
    } else {
        conv_epilogue<T, TM, TP, 0, true>(p, acc, n0 + wm * (TM * 16) + fq * CH, m0 + wn * (TP * 16), fr);
    }
#ifdef YOLO_EXPERIMENT
    if (p.trace && tid == 0) {          // YOLO_CONV_TRACE: phase timestamps (100 MHz) + placement of wave 0 of every block
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *r = p.trace + (size_t)raw_x * 8;
        r[0] = t_start; r[1] = t_setup; r[2] = t_loop; r[3] = wall_clock64();
        r[4] = __builtin_amdgcn_s_getreg(0xF804);      // HW_ID
        r[5] = __builtin_amdgcn_s_getreg(0xF814);      // XCC_ID
        r[6] = t_first;                                // first K tile landed (prologue DMA latency)
        r[7] = (unsigned long long)clock64() - c_start;
    }
#endif
}

}  // namespace yolo
