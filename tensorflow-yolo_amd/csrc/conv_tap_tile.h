// The tap-reuse 3x3 tile of conv_tap.hip as a device function (one workgroup, one tile), shared by its kernel there and by the chained launch
// of conv_chain.hip.  See conv_tap.hip for the design; the compile-time switches of the K loop live here.
#pragma once
#include "conv_common.h"
#include <type_traits>

#ifndef YOLO_TAP_ASM_MFMA
#define YOLO_TAP_ASM_MFMA 1
#endif
#ifndef YOLO_TAP_W_FIRST
#define YOLO_TAP_W_FIRST 1
#endif
#ifndef YOLO_TAP_AH_ALL
#define YOLO_TAP_AH_ALL 1
#endif
#ifndef YOLO_TAP_PIL_RECOMPUTE
#define YOLO_TAP_PIL_RECOMPUTE 0
#endif
#ifndef YOLO_TAP_PIL
#define YOLO_TAP_PIL 1
#endif
#ifndef YOLO_TAP_DBG        // timing experiments of tools/ (make EXTRA=-DYOLO_TAP_DBG=..): never set in the product build
#define YOLO_TAP_DBG 0
#endif
#ifndef YOLO_TAP_RECOMPUTE_ADDR
#define YOLO_TAP_RECOMPUTE_ADDR 1
#endif
#ifndef YOLO_TAP_LATE_FROM
#define YOLO_TAP_LATE_FROM 4
#endif
#ifndef YOLO_TAP_PRIO       // experiment (profiles/r05_ablation.md): s_setprio around the MFMA burst of a tap
#define YOLO_TAP_PRIO 0
#endif
#ifndef YOLO_TAP_STAGGER
#define YOLO_TAP_STAGGER 1
#endif

namespace yolo {

namespace {

typedef __attribute__((address_space(3))) void tap_lds_void;

// voff: per-lane byte offset (loop invariant, range-checked: an invalid offset writes zeros); soff: wave-uniform
// byte offset of the K position, added by the hardware after the range check -> no per-tap address registers.
// AUX: cache policy of the load (0 plain; 16 = sc1: served by the XCD's L2, never by this CU's L1 -- conv_chain.hip).
template <int AUX = 0>
__device__ __forceinline__ void tap_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char *lds_dst, uint32_t voff, uint32_t soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (tap_lds_void *)lds_dst, 16, voff, soff, 0, AUX);
#else
    (void)rsrc; (void)lds_dst; (void)voff; (void)soff;
#endif
}

// acc += a . b IN PLACE (D = C).  hipcc's own MFMAs (the builtin) get an untied destination whenever its allocator finds that cheaper
// locally, and in these fully unrolled K loops the 16 accumulators of a wave then wander through ~20 spare registers (`v_mfma v[38:41], a, b,
// v[66:69]`): the 128-register tiles spilled patch-DMA offsets for it (reloads behind `vmcnt(0)` inside the loop) and took their weight fragments
// one at a time.  As an asm statement with a read-write operand the accumulator stays where it is: the dominant tile needs 107 registers
// instead of 128.  What hipcc does not do for an asm statement (cdna_hip_programming.md 5.7): pad its hazards -- the accumulate chain
// MFMA -> MFMA on the same D = C needs none, A / B come from LDS reads behind hipcc's own `s_waitcnt`, and tap_mfma_drain() stands between the
// last MFMA and the epilogue's vector reads of the accumulators.
template <typename T>
__device__ __forceinline__ void tap_mfma(float4v &acc, const uint4v &a, const uint4v &b) {
#if YOLO_TAP_ASM_MFMA && defined(__HIP_DEVICE_COMPILE__)
    if constexpr (sizeof(T) == 2) {
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        return;
    }
#endif
    acc = mma_chunk<T>(a, b, acc);
}

// between the K loop's last in-place MFMA and the first reader of an accumulator that is not an MFMA: the wait states hipcc would have
// inserted for its own MFMAs (XDL write -> VALU read, at most 18 for this opcode class), and every accumulator made opaque BEHIND them
// (volatile statements keep their order), so that no consumer is scheduled above
template <int TM, int TP>
__device__ __forceinline__ void tap_mfma_drain(float4v (&acc)[TM][TP]) {
#if YOLO_TAP_ASM_MFMA && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0][0]));
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) asm volatile("" : "+v"(acc[a][b]));
#endif
}

template <int N>
__device__ __forceinline__ void tap_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ int tap_swz_w(int r) { return (0x78 >> (2 * ((r >> 2) & 3))) & 3; }   // weight tile: {0,2,3,1}[(r>>2)&3]

constexpr int kTapStreamBiasFloats = 512;      // stream kernel: couts whose bias fits its LDS copy (2 KiB)

}  // namespace

// 8 waves = WM x WN; a wave owns TM*16 couts x TP*16 positions; PRG = 16-row groups of one patch buffer.
// MODE 1: padded-linear positions (above).  MODE 2: 2-D tiles for wide maps, where 2W+4 halo positions would not fit:
// a block owns TH x 16 output pixels (TH = NB / 16, fragment = one tile row), the patch is (TH + 2) rows of PW = 24
// slots (18 used: PW a multiple of 8 keeps the swizzle phase of every fragment row equal, so fragment offsets stay
// immediates), tap shift kh * 24 + kw; pixels of partial tiles outside the image are computed and dropped.
// T = _Float16 (32-channel slices, mfma_f32_16x16x32_f16) or float (16-channel slices, four mfma_f32_16x16x4f32 per
// fragment pair: the exact fp32 FMA chain of conv.hip) -- the LDS geometry is in 16-byte chunks either way.
// (F32 instead of the element type as template parameter: rocprofv3 does not demangle `_Float16` template arguments, and
// yolo_kernel_info.symbol must be the name its kernel trace prints)
// SPLITK: the split-K instantiation (blockIdx.y = K split, raw float32 partial sums out; 128 x 128 tile only).  A template
// parameter, not a run-time branch: with the branch in the code the register allocation of the big tiles changed (46-64
// VGPRs spilled, scratch traffic doubling the kernel's HBM writes: profiles/r02_ablation.md).
// FAST: the lean epilogue of conv_common.h (conv_epilogue_fast) instead of the generic one -- an instantiation of its own, not a
// run-time branch: with both epilogues in one kernel the 128-register tiles spilled two patch-DMA offsets, reloaded inside the K
// loop behind a vmcnt(0) that drains the DMA queue.  fp16, whole K, MODE 1 / 2.
// FUSE2: FAST + the 1x1 conv behind this one computed by the same workgroups (conv_common.h: conv_epilogue_fused_1x1); the
// 128 x 256 2-D tile only (all 128 couts of 256 positions in one workgroup).
// The kernel is one call of conv3x3_tap_tile: the workgroup's whole life for ONE tile, on LDS the caller owns -- so that the chained launch
// (conv_chain.hip: tiles of consecutive layers handed out in dependency order to resident workgroups) runs the same code.
//   bid: tile index of the launch (already through xcd_remap); raw_x: the workgroup's own index (ticket / slab of the in-launch K split);
//   block_y / nsplit_y: K split and number of splits (0 / 1 without);
//   COH: activations (patch, residual) are read with sc1 loads -- they may have been written by another CU of this XCD in THIS launch.
template <int WM, int TM, int TP, int PRG, bool FUSE2>
constexpr int conv3x3_tap_lds_bytes() {
    constexpr int ring = 3 * (WM * TM * 16) * 64 + 2 * PRG * 1024;
    return FUSE2 && kFuse2LdsBytes > ring ? kFuse2LdsBytes : ring;
}
template <bool F32, int WM, int WN, int TM, int TP, int PRG, int OCC, int MODE, bool SPLITK, bool FAST = false, bool FUSE2 = false, bool COH = false, bool SC1 = COH, bool PAIR = false>
__device__ __forceinline__ void conv3x3_tap_tile(const ConvParams p, unsigned char *const smem, const int bid, const int raw_x, const int block_y,
                                                 const int nsplit_y, const int pair_half = 0, const bool tile_valid = true) {
    typedef typename std::conditional<F32, float, _Float16>::type T;
    constexpr int LDAUX = SC1 ? 16 : 0;
    constexpr bool TWO_D = MODE == 2 || MODE == 3;       // MODE 3: the 2-D tiles of MODE 2 with the max-pool behind the conv taken in the epilogue
    constexpr bool S2 = MODE == 4;          // 3x3 / stride 2 over the four parity planes of the input (see run_slice_s2 below)
    constexpr int PADQ = TWO_D ? 2 : 1;
    constexpr int NW = WM * WN;
    constexpr int S = 3;                    // weight ring slots (9 taps per slice: slot = tap % 3)
    constexpr bool STAG = YOLO_TAP_STAGGER != 0 && !F32 && !SPLITK && TM % 2 == 0 && NW == 8 && MODE == 4 && OCC == 4;    // half-tap stagger of waves 4-7 (below)
    constexpr int TMH = TM / 2;
    // POSITION-INTERLEAVED FRAGMENTS (round 5; padded-linear fp16 tiles).  profiles/r05_ablation.md section 1: the K loop is bound by the fragment
    // traffic LDS -> registers (without the reads a launch is 18-22 % shorter; with six per 16 MFMAs instead of eight 7-14 %).  The three taps
    // of a kernel row read the SAME patch rows shifted by one position.  With fragment b of a wave = the 16 positions w0 + 16 b + fr
    // (lane row fr) a shift by one position moves data across lanes; with fragment b = the positions w0 + TP fr + b it moves data to the NEXT
    // FRAGMENT: tap kw of output fragment b needs the input positions w0 + TP fr + (b + kw) = input fragment j = b + kw, and j = TP, TP + 1 are
    // fragments 0, 1 one lane row further (positions w0 + TP (fr + 1) + ...): TP + 2 fragment reads serve the 3 TP fragment uses of a kernel
    // row.  12 -> 6 position reads per row at TP = 4: 18 fragment reads per 48 MFMAs instead of 24, no extra registers (fragment j lives in
    // register set j mod TP; TP and TP + 1 are read into the sets of fragments 0 and 1 when tap kw = 0 / 1 has issued its last MFMA on
    // them), no vector work.  For a fragment to be 16 CONSECUTIVE LDS rows the patch is stored de-interleaved: patch position R lives in
    // plane R mod TP at row R / TP (planes of PL rows) -- free, because the LDS-DMA takes a per-lane source offset anyway.  The epilogue's
    // lane <-> pixel map changes accordingly (conv_common.h: frag_pos).
    constexpr bool PIL = YOLO_TAP_PIL != 0 && MODE == 1 && !F32;
    constexpr int PL = (PRG * 16) / TP;             // patch rows per plane
    constexpr int ROWB = 64;
    constexpr int NA = WM * TM * 16;
    constexpr int NB = WN * TP * 16;
    constexpr int JA_TOT = NA / 16;         // weight DMA wave-instructions per tap
    constexpr int JA = (JA_TOT + NW - 1) / NW;      // per wave (a 64-cout tile has 4: waves 4..7 carry none)
    // MODE 2: patch row pitch in positions.  24 (a multiple of 8) keeps the swizzle phase of every fragment row equal, so the
    // fragment offsets are immediates; the three-workgroups-per-CU tile (OCC 6) takes the minimal pitch 18 to fit its two patch
    // buffers into a third of the LDS and pays one address computation per fragment and tap instead
    constexpr int PW = (TWO_D && OCC >= 6) ? 18 : 24;
    constexpr int FROW = TWO_D ? PW : 16;       // patch rows between consecutive fragments of a wave
    constexpr int JP = (PRG + NW - 1) / NW; // patch DMA wave-instructions per wave per slice
    constexpr int CH = 4 * TM;
    constexpr int A_BYTES = NA * ROWB;
    constexpr int P_BYTES = PRG * 1024;
    static_assert(NW == 8, "eight waves");
    static_assert(JA_TOT % NW == 0 || JA_TOT < NW, "weight tile must split evenly over the waves (or be smaller than them)");
    static_assert(!TWO_D || PRG * 16 >= (NB / 16 + 2) * PW, "patch buffer too small for the 2-D tile");
    if constexpr (F32 && (TP > 2 || OCC >= 6)) return;     // never launched (launch_conv_tap refuses): no registers for the second accumulator
    if constexpr (SPLITK && !(WM == 2 && WN == 4 && TM == 4 && (TP == 2 || ((TP == 4 || TP == 3) && !F32)) && MODE == 1)) return;    // split-K: the 128 x 128 tile; 128 x 256 and 128 x 192 (fp16) for the in-launch pair
    if constexpr (MODE == 3 && (TP % 2 != 0 || SPLITK)) return;      // split-K: the 128 x 128 tile only
    static_assert(!FAST || (!F32 && !SPLITK && MODE != 3), "the lean epilogue: fp16, whole K, plain output");
    static_assert(!S2 || JA_TOT % NW == 0, "stride 2: every wave carries weights");
    if constexpr (S2 && (F32 || SPLITK)) return;            // never launched (launch_conv_tap refuses): fp16, whole K only
    static_assert(!FUSE2 || (FAST && WM == 2 && WN == 4 && TM == 4 && TP == 4), "back-to-back 1x1: the 128 x 256 tiles (eight waves)");
    constexpr int LDS_BYTES = conv3x3_tap_lds_bytes<WM, TM, TP, PRG, FUSE2>();
    static_assert(LDS_BYTES >= S * A_BYTES + 2 * P_BYTES, "LDS size of the tile");
    unsigned char *const smemP = smem + S * A_BYTES;

    // PAIR (conv3x3_tap_pair_kernel): this is one HALF (eight waves, its own LDS) of a 16-wave workgroup; every s_barrier below then meets all sixteen
    // waves, and the two halves run half a tap apart (see the kernel)
    int tid_ = PAIR ? (int)(threadIdx.x & 511u) : (int)threadIdx.x;
    // (the chained launch calls this in a loop: without the empty statement everything derived from the thread index is loop-invariant, gets
    // hoisted and stays live across the whole iteration -- +24 scalar and ~9 vector registers, 6 spilled)
    if constexpr (COH) asm volatile("" : "+v"(tid_));
    const int tid = tid_;
#ifdef YOLO_EXPERIMENT      // block trace (tools/trace_blocks.py); not in the product build
    const unsigned long long t_start = p.trace ? wall_clock64() : 0ull;
    const unsigned long long c_start = p.trace ? (unsigned long long)clock64() : 0ull;
    // experiment (tools/offset_probe.py): the workgroups with an odd threadgroup id on their CU start (dbg >> 8) x 0.25 us late
    if (!PAIR && (p.dbg >> 8) > 0 && ((__builtin_amdgcn_s_getreg(0xF804) >> 16) & 1u)) {
        const unsigned long long until = wall_clock64() + (unsigned long long)(p.dbg >> 8) * 25ull;
        while (wall_clock64() < until) __builtin_amdgcn_s_sleep(4);
    }
#endif
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int mt = (int)fdiv((uint32_t)bid, p.dtiles_n);
    const int nt = bid - mt * p.n_tiles_n;
    const int n0 = nt * NA;
    const int q0 = mt * p.q_stride;    // NB, or (H+1)(W+1) for the image-aligned tile (variant 9)
    const bool has_a = JA_TOT % NW == 0 || wave < JA_TOT;   // wave-uniform

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wgt), 0, p.wgt_bytes, 0x00020000);

    // ---- DMA geometry ----------------------------------------------------------------------------
    const int lrow = lane >> 2;
    uint32_t a_off[JA];
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int r = (j * NW + wave) * 16 + lrow;              // LDS row of the weight tile
        const int ws = r / (TM * 16), R = r % (TM * 16);
        const int tm = R >> 4, g4 = (R >> 2) & 3, jj = R & 3;
        const int ch = ws * (TM * 16) + g4 * CH + 4 * tm + jj;  // the cout that row holds (see conv_epilogue)
        a_off[j] = (uint32_t)(n0 + ch) * p.wrow_bytes + (uint32_t)(((lane & 3) ^ tap_swz_w(lrow)) << 4);
    }
    // channel slices of this workgroup: all of them, or one K split's share (split-K, blockIdx.y)
    const int c_begin = SPLITK ? block_y * p.kunits : 0;
    const int C = SPLITK ? ((c_begin + p.kunits < (p.cin_chunks >> 2)) ? c_begin + p.kunits : (p.cin_chunks >> 2)) : (p.cin_chunks >> 2);
    const int KT = 9 * C;
    auto issue_weights = [&](int tap, int c, int slot) {
        const uint32_t ka = (uint32_t)(tap * p.cin_chunks + 4 * c) * 16;
        if (has_a) {
#pragma unroll
            for (int j = 0; j < JA; ++j) tap_dma16(rs_w, smem + slot * A_BYTES + (j * NW + wave) * 1024, a_off[j], ka);
        }
    };
    // (PIL tiles: the first two weight tiles are requested HERE, in front of the patch geometry -- two multiply-shift divisions per patch row
    // group and lane, ~0.5 us of the 1 us a workgroup spends in setup -- so that their latency runs under it; DMA order W0, W1, patch)
    constexpr bool W_FIRST = YOLO_TAP_W_FIRST != 0 && YOLO_TAP_PIL != 0 && MODE == 1 && !F32;
    if constexpr (W_FIRST) {
        issue_weights(0, c_begin, 0);
        issue_weights(1, c_begin, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // patch row R <-> position q0 - (W+2) + R; row group g = j NW + wave
    // (the last of a wave's JP row groups may lie beyond the patch: waves >= JP_FULL issue one instruction fewer)
    constexpr int JP_FULL = PRG - (JP - 1) * NW;    // waves that own JP row groups
    const bool jp_full = wave < JP_FULL;            // wave-uniform
    uint32_t b_off[JP];
    const uint32_t csw_p = (uint32_t)(((lane & 3) ^ (((lrow >> 2) & 1) << 1)) << 4);
    int t2_n = 0, t2_y0 = 0, t2_x0 = 0;     // MODE 2: image and first output pixel of this block's tile
    if (TWO_D) {
        t2_n = (int)fdiv((uint32_t)mt, p.dqHW);             // qHW = tiles per image, qW = tiles per tile row
        const int r = mt - t2_n * p.qHW;
        const int ty = (int)fdiv((uint32_t)r, p.dqW);
        t2_y0 = ty * (NB / 16);
        t2_x0 = (r - ty * p.qW) * 16;
    }
#pragma unroll
    for (int j = 0; j < JP; ++j) {
        const int g = j * NW + wave;
        bool ok;
        int n, y, x;
        if (TWO_D) {
            const int R = g * 16 + lrow;
            const int pr = R / PW, pc = R - pr * PW;
            n = t2_n; y = t2_y0 - 1 + pr; x = t2_x0 - 1 + pc;
            ok = g < PRG && pc < 18 && pr < NB / 16 + 2 && t2_n * p.HoWo < p.M && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        } else {
            int Rp = g * 16 + lrow;                 // LDS row of the patch buffer -> patch position
            bool in_planes = true;
            if constexpr (PIL) {
                const int plane = Rp / PL, idx = Rp - plane * PL;
                in_planes = plane < TP;             // (PRG * 16 rows need not be a multiple of TP)
                Rp = idx * TP + plane;
            }
            const int q = q0 - (p.qW + 1) + Rp;
            ok = g < PRG && in_planes && q >= 0 && q < p.Mq;
            const int qq = ok ? q : 0;
            n = (int)fdiv((uint32_t)qq, p.dqHW);
            const int r = qq - n * p.qHW;
            y = (int)fdiv((uint32_t)r, p.dqW);
            x = r - y * p.qW;
            if (S2) { ok = ok && x < p.Wo && y < p.Ho; y *= 2; x *= 2; }     // plane (0, 0) of the input: pixel (2 y', 2 x')
            else ok = ok && x < p.W && y < p.H;
        }
        const long long e = (long long)n * p.in_img_stride + ((long long)y * p.W + x) * p.in_ld + p.in_coff;
        b_off[j] = ok ? (uint32_t)(e * (long long)sizeof(T)) + csw_p : YOLO_INVALID_OFF;
    }

    auto issue_patch = [&](int c, int buf, uint32_t plane_off = 0u) {
        const uint32_t koff = (uint32_t)c * ROWB + plane_off;
#pragma unroll
        for (int j = 0; j < JP; ++j)
            if (j + 1 < JP || jp_full) tap_dma16<LDAUX>(rs_in, smemP + buf * P_BYTES + (j * NW + wave) * 1024, b_off[j], koff);
    };

    float4v acc[TM][TP];
    // float32: second-level accumulator (conv_common.h: flush_acc); only the TP <= 2 tiles have the registers for it at
    // two workgroups per CU, so those are the float32 tiles (kTapF32)
    float4v acc2[F32 ? TM : 1][F32 ? TP : 1];
    const int fr = lane & 15, fq = lane >> 4;
    // the accumulators start from the bias (conv_common.h: conv_init_acc_bias); split-K partial sums carry none
    // (called at the head of each of the two K-loop forms below -- early / late waves -- rather than once in front of the branch: the
    // compiler otherwise keeps the 16 bias registers alive across the first form's loop to initialise the second's accumulators)
    auto init_acc = [&]() {
        if constexpr (SPLITK) {
            if ((TP != 2 || p.pair) && block_y == 0) {    // in-launch pair: the bias rides in half 0
                conv_init_acc_bias<TM, TP>(p, acc, n0 + wm * (TM * 16) + fq * CH);
            } else {
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TP; ++b) acc[a][b] = float4v{0.f, 0.f, 0.f, 0.f};
            }
        } else {
            conv_init_acc_bias<TM, TP>(p, acc, n0 + wm * (TM * 16) + fq * CH);
        }
    };
    if constexpr (!STAG) init_acc();
    if constexpr (F32) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TP; ++b) acc2[F32 ? a : 0][F32 ? b : 0] = float4v{0.f, 0.f, 0.f, 0.f};
    }

    const int a_frag = (wm * TM * 16 + fr) * ROWB + (((fq ^ tap_swz_w(fr)) & 3) << 4);
    const int rb = wn * TP * FROW + fr;     // patch row of this lane's position for tap (0, 0)

    auto compute = [&](int slot, int buf, int shift) {
        const unsigned char *A = smem + slot * A_BYTES + a_frag;
        if constexpr (!TWO_D && YOLO_TAP_RECOMPUTE_ADDR && STAG) asm volatile("" : "+s"(shift));
        const int R = rb + shift;
        const unsigned char *B = smemP + buf * P_BYTES + (R << 6) + ((fq << 4) ^ ((R & 4) << 3));
        uint4v fa[TM], fb[TP];
        if constexpr ((YOLO_TAP_DBG & 64) != 0) {       // timing experiment (results wrong): no fragment reads (whatever the registers hold)
#pragma unroll
            for (int a = 0; a < TM; ++a) asm volatile("" : "=v"(fa[a]));
#pragma unroll
            for (int b = 0; b < TP; ++b) asm volatile("" : "=v"(fb[b]));
        } else {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            if constexpr ((YOLO_TAP_DBG & 128) != 0) { if (a == TM - 1) { asm volatile("" : "=v"(fa[a])); continue; } }     // timing experiment: 6 reads per 16 MFMAs
            fa[a] = *reinterpret_cast<const uint4v *>(A + a * 16 * ROWB);
        }
#pragma unroll
        for (int b = 0; b < TP; ++b) {
            if constexpr ((YOLO_TAP_DBG & 128) != 0) { if (b == TP - 1) { asm volatile("" : "=v"(fb[b])); continue; } }
            if constexpr (TWO_D && (FROW & 7) != 0) {       // the swizzle phase differs from fragment row to fragment row
                const int Rb = R + b * FROW;
                fb[b] = *reinterpret_cast<const uint4v *>(smemP + buf * P_BYTES + (Rb << 6) + ((fq << 4) ^ ((Rb & 4) << 3)));
            } else {
                fb[b] = *reinterpret_cast<const uint4v *>(B + b * FROW * ROWB);
            }
        }
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TP; ++b) tap_mfma<T>(acc[a][b], fa[a], fb[b]);
    };

    // position-interleaved fragments (PIL, above): input fragment j of kernel row kh = patch positions u + TP fr, u = kh qW + wave offset + j
    uint4v G[PIL ? TP : 1];
    auto g_ptr = [&](int buf, int u) {
        // (3 (TP + 2) loop-invariant addresses per patch buffer: with the in-place MFMAs the 128-register tiles have the registers to keep
        // them -- 18 at TP = 4 -- and computing them where they are used cost +35 % vector instructions, +5 % wave cycles: profiles/r05_ablation.md)
        if constexpr (YOLO_TAP_PIL_RECOMPUTE != 0) asm volatile("" : "+s"(u));
        const int plane = (int)((unsigned)u % (unsigned)TP), idx0 = (int)((unsigned)u / (unsigned)TP);
        const int L = plane * PL + idx0 + fr;
        return smemP + buf * P_BYTES + (L << 6) + ((fq << 4) ^ ((L & 4) << 3));
    };
    // The position fragments a tap needs are requested at the END of the tap in front of it, i.e. BEFORE the barrier between them: they
    // read the PATCH, which is complete and visible a slice ahead and which no DMA overwrites before the slice after next, so they may be
    // in flight across the barrier -- behind it only the weight fragments (whose DMA the barrier publishes) are waited for.  After tap
    // kw = 0 / 1: fragment j = TP + kw into the registers of fragment j = kw (dead); after kw = 2: the TP fragments of the next kernel
    // row (next slice: the other patch buffer).  `nbuf`: the buffer the tap behind this one reads.
    auto compute_pil = [&](int slot, int buf, int kh, int kw, int nbuf, auto &&issue_dma) {
        if constexpr (PIL) {
            const unsigned char *A = smem + slot * A_BYTES + a_frag;
            const int u0 = kh * p.qW + wn * (TP * 16);
            constexpr int AH = YOLO_TAP_AH_ALL ? TM : (TM >= 4 ? TM / 2 : TM);        // weight fragments in flight at a time (two register sets of TM / 2 at TM = 4, 8)
            uint4v fa[TM];
#pragma unroll
            for (int a0 = 0; a0 < TM; a0 += AH) {
#pragma unroll
                for (int a = a0; a < a0 + AH; ++a) fa[a] = *reinterpret_cast<const uint4v *>(A + a * 16 * ROWB);
                // the tap's DMA requests go out BEHIND the first weight-fragment reads: the reads' latency then covers the requests' issue
                // (~60-180 cycles each) instead of standing behind it -- nothing else is between the barrier and the tap's first MFMA
                if (a0 == 0) issue_dma();
                if constexpr (YOLO_TAP_PRIO == 1) __builtin_amdgcn_s_setprio(3);      // experiment: a wave's MFMA burst is not pre-empted by the other workgroup's
                if constexpr (YOLO_TAP_PRIO == 2) __builtin_amdgcn_s_setprio(0);      // experiment: the reverse (reads / DMA requests first)
#pragma unroll
                for (int a = a0; a < a0 + AH; ++a) {
                    if constexpr (PAIR) {
                        if (a == TM / 2) {      // the half-tap point: the other half of the workgroup is at its tap boundary (and the other way round)
                            __builtin_amdgcn_sched_barrier(0);
                            __builtin_amdgcn_s_barrier();
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
#pragma unroll
                    for (int b = 0; b < TP; ++b) tap_mfma<T>(acc[a][b], fa[a], G[(b + kw) % TP]);
                }
                if constexpr (YOLO_TAP_PRIO == 1) __builtin_amdgcn_s_setprio(0);
                if constexpr (YOLO_TAP_PRIO == 2) __builtin_amdgcn_s_setprio(3);
            }
            if (kw < 2) G[kw % TP] = *reinterpret_cast<const uint4v *>(g_ptr(buf, u0 + TP + kw));
            else {
                const int un = (kh < 2 ? (kh + 1) * p.qW : 0) + wn * (TP * 16);
#pragma unroll
                for (int b = 0; b < TP; ++b) G[b] = *reinterpret_cast<const uint4v *>(g_ptr(nbuf, un + b));
            }
        }
    };

    // HALF-TAP STAGGER (round 5).  All eight waves of a workgroup run the same program between the same barriers, so the two waves
    // that share a SIMD (w and w + 4) reach their fragment reads -- and the ~150-250 cycles until the first of them is back --
    // together, and nothing feeds the matrix pipe meanwhile: with the barriers removed (wrong results; profiles/r05_ablation.md)
    // the launches ran 18-20 % faster just because the waves drift apart.  Waves 4-7 ("late") therefore run HALF a tap behind:
    // between barrier t and t + 1 they first issue the second half (cout fragments TM/2 ..) of tap t - 1 from fragments they read
    // BEFORE barrier t and kept in registers (the same registers the tap's reads use anyway), then read and issue the first half of
    // tap t, then read the second half's weight fragments and hold them across barrier t + 1.  The matrix pipe has their 8 MFMAs
    // to run while the early waves wait for their reads, and the early waves' MFMAs while the late waves wait for theirs.  Every
    // accumulator still sums its taps in the same order (bit-identical results); the late waves' reads of a ring slot are
    // complete (lgkmcnt(0)) before the barrier behind which another wave's DMA may overwrite it; DMA issue and vmcnt waits are
    // the same for both kinds.  fp16, whole K.
    // Measured (interleaved A/B, profiles/r05_ablation.md): the two-per-CU stride-2 tiles -7 %; every stride-1 tile and the one-per-CU
    // tiles +2 ... +5 % SLOWER (what bounds them is the fragment traffic LDS -> registers itself, not the wait for it) -- so only there
    // (STAG, defined with the tile constants above).
    uint4v ha[STAG ? TMH : 1], hb[STAG ? TP : 1];       // late waves: the held fragments (second-half weights, positions) of the previous tap
    auto late_finish = [&]() {
        if constexpr (STAG) {
#pragma unroll
            for (int a = 0; a < TMH; ++a)
#pragma unroll
                for (int b = 0; b < TP; ++b) tap_mfma<T>(acc[TMH + a][b], ha[a], hb[b]);
        }
    };
    auto compute_late = [&](int slot, int buf, int shift) {
        if constexpr (STAG) {
            late_finish();
            __builtin_amdgcn_sched_barrier(0);      // (the reads below reuse the registers of the fragments just consumed: none is hoisted above)
            const unsigned char *A = smem + slot * A_BYTES + a_frag;
            if constexpr (!TWO_D && YOLO_TAP_RECOMPUTE_ADDR) asm volatile("" : "+s"(shift));     // opaque: the patch address is computed per tap (five vector
            // instructions beside 16 MFMAs) instead of living in nine loop-invariant registers the 128-register tiles do not have
            const int R = rb + shift;
            const unsigned char *B = smemP + buf * P_BYTES + (R << 6) + ((fq << 4) ^ ((R & 4) << 3));
            uint4v fa[TMH];
#pragma unroll
            for (int a = 0; a < TMH; ++a) fa[a] = *reinterpret_cast<const uint4v *>(A + a * 16 * ROWB);
#pragma unroll
            for (int b = 0; b < TP; ++b) {
                if constexpr (TWO_D && (FROW & 7) != 0) {
                    const int Rb = R + b * FROW;
                    hb[b] = *reinterpret_cast<const uint4v *>(smemP + buf * P_BYTES + (Rb << 6) + ((fq << 4) ^ ((Rb & 4) << 3)));
                } else {
                    hb[b] = *reinterpret_cast<const uint4v *>(B + b * FROW * ROWB);
                }
            }
#pragma unroll
            for (int a = 0; a < TMH; ++a)
#pragma unroll
                for (int b = 0; b < TP; ++b) tap_mfma<T>(acc[a][b], fa[a], hb[b]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < TMH; ++a) ha[a] = *reinterpret_cast<const uint4v *>(A + (TMH + a) * 16 * ROWB);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slot may be overwritten behind the next barrier
        }
    };
    const bool late = STAG && wave >= YOLO_TAP_LATE_FROM;        // wave-uniform
    // ---- prologue: patch of slice 0, weights of taps 0 and 1 --------------------------------------
#ifdef YOLO_EXPERIMENT
    const unsigned long long t_setup = p.trace ? wall_clock64() : 0ull;
#endif
    // STRIDE 2 (MODE 4).  Output position (y, x) reads input pixel (2y + kh - 1, 2x + kw - 1): row 2y - 1 / 2y / 2y + 1 lies in the
    // row-parity plane 1 / 0 / 1 at plane row y - 1 / y / y (columns alike), so over the four parity planes P[py][px](y', x') =
    // in(2y' + py, 2x' + px) -- each of the OUTPUT's size -- the conv is nine (plane, shift) pairs with shifts dy, dx in {-1, 0}:
    // plane (1,1) four taps, (1,0) and (0,1) two each, (0,0) one.  The positions walk the padded-linear grid of the OUTPUT map
    // (qW = Wo + 1), a patch is NB + Wo + 2 positions of ONE plane (row R <-> position q0 - (Wo + 2) + R, tap shift
    // (dy + 1) qW + (dx + 1)), gathered by the same per-lane LDS-DMA offsets for every plane (a pad position is a pad position in
    // all four; the plane's pixel offset (py W + px) in_ld rides in the DMA's scalar offset).  Per 32-channel slice the tap order is
    // A A D A A B B C C (A = plane (1,1), D = (0,0), B = (1,0), C = (0,1)) over TWO patch buffers X, Y that swap roles from slice
    // to slice: A in X; D -> Y requested at step 0 (Y held the last slice's C), B -> Y at step 3 (after D's only tap), C -> X at
    // step 5 (after A's last), the next slice's A -> Y at step 7 (after B's last): every patch is requested two taps before its
    // first use, like the weights.  4 x 21 KiB of patch + 72 KiB of weights per slice and 128 x 256 tile against 9 x 16 + 72 KiB
    // for the per-tap gather of conv_dma.hip -- and two workgroups per CU.
    const uint32_t pl01 = S2 ? (uint32_t)p.in_ld * 2u : 0u, pl10 = S2 ? (uint32_t)(p.W * p.in_ld) * 2u : 0u;
    if constexpr (S2) {
        issue_patch(c_begin, 0, pl10 + pl01);
        issue_weights(0, c_begin, 0);
        issue_weights(2, c_begin, 1);
    } else if constexpr (W_FIRST) {
        issue_patch(c_begin, 0);        // (the weights are on their way: above)
    } else {
        issue_patch(c_begin, 0);
        issue_weights(0, c_begin, 0);
        issue_weights(1, c_begin, 1);
    }
    auto run_slice_s2 = [&](int c, auto bufc, auto latec) {
        constexpr int X = decltype(bufc)::value, Y = X ^ 1;
        constexpr bool LATE = decltype(latec)::value;
        constexpr int kTapOf[9] = {0, 2, 4, 6, 8, 1, 7, 3, 5};          // kh * 3 + kw of step s
        constexpr int kBufOf[9] = {X, X, Y, X, X, Y, Y, X, X};
        constexpr int kShOf[9] = {0, 1, 3, 2, 3, 1, 3, 2, 3};           // shift code: bit 1 = + qW (dy = 0), bit 0 = + 1 (dx = 0)
        const bool more = c + 1 < C;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            // in flight behind what this step needs: the weights of step s + 1 and the patch requested at step s - 1 (if any)
            const bool last = !more && s == 8;
            const bool patch_prev = s == 1 || s == 4 || s == 6 || (s == 8 && more);
            if (last) tap_wait_vm<0>();
            else if (!patch_prev) tap_wait_vm<JA>();
            else if (jp_full) tap_wait_vm<JA + JP>();
            else tap_wait_vm<JA + JP - 1>();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            {   // weights two steps ahead
                const int s2 = s + 2 < 9 ? s + 2 : s + 2 - 9;
                const int c2 = s + 2 < 9 ? c : c + 1;
                if (c2 < C) issue_weights(kTapOf[s2], c2, (s + 2) % S);
            }
            if (s == 0) issue_patch(c, Y, 0u);
            if (s == 3) issue_patch(c, Y, pl10);
            if (s == 5) issue_patch(c, X, pl01);
            if (s == 7 && more) issue_patch(c + 1, Y, pl10 + pl01);
            if constexpr (LATE) compute_late(s % S, kBufOf[s], ((kShOf[s] & 2) ? p.qW : 0) + (kShOf[s] & 1));
            else compute(s % S, kBufOf[s], ((kShOf[s] & 2) ? p.qW : 0) + (kShOf[s] & 1));
        }
    };

    // one 32-channel slice; the patch buffer index is a compile-time constant (LDS immediates, no address registers)
    auto run_slice = [&](int c, auto bufc, auto latec) {
        constexpr int buf = decltype(bufc)::value;
        constexpr bool LATE = decltype(latec)::value;
        const bool more = c + 1 < C;        // a next slice exists: its patch is fetched during this one
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // Wait for the weights of this tap.  Issue order per tap: weights(tap+2), then (tap 0 only) the next
            // patch; younger than weights(tap) are weights(tap+1) and, at taps 1 and 2, that patch.
            const bool last = !more && tap == 8;
            const bool with_patch = more && (tap == 1 || tap == 2);
            if constexpr ((YOLO_TAP_DBG & 32) != 0) { if (last) tap_wait_vm<0>(); } else     // timing experiment (results wrong): no DMA waits inside the K loop
            if constexpr ((YOLO_TAP_DBG & 256) != 0) {      // timing experiment (results right): the weights of tap+1 must have landed as well
                if (with_patch && tap == 1) { if (jp_full) tap_wait_vm<JP>(); else tap_wait_vm<JP - 1>(); }
                else tap_wait_vm<0>();
            } else
            if (last) tap_wait_vm<0>();
            else if (has_a) {
                if (!with_patch) tap_wait_vm<JA>();
                else if (jp_full) tap_wait_vm<JA + JP>();
                else tap_wait_vm<JA + JP - 1>();
            } else {            // this wave issues patch instructions only
                if (!with_patch) tap_wait_vm<0>();
                else if (jp_full) tap_wait_vm<JP>();
                else tap_wait_vm<JP - 1>();
            }
            // Nothing is scheduled across the barrier: every ds_read of this tap is consumed by an MFMA before the wave
            // arrives, so a slot is provably idle when another wave's DMA (issued after the barrier) overwrites it.
            // (The compiler otherwise sinks the last fragment reads + MFMAs below the barrier: 0.5 % faster, but safe
            // only by timing.)
            __builtin_amdgcn_sched_barrier(0);
            // (timing experiments, results wrong: YOLO_TAP_DBG 8 = a barrier every third tap only, 16 = none)
            if constexpr (!((YOLO_TAP_DBG & 16) != 0 || ((YOLO_TAP_DBG & 8) != 0 && tap % 3 != 0))) __builtin_amdgcn_s_barrier();
            auto issue_dma = [&]() {
                {   // weights two taps ahead
                    const int t2 = tap + 2 < 9 ? tap + 2 : tap + 2 - 9;
                    const int c2 = tap + 2 < 9 ? c : c + 1;
                    if (c2 < C) issue_weights(t2, c2, (tap + 2) % S);
                }
                if (tap == 0 && more) issue_patch(c + 1, buf ^ 1);
            };
            const int kh = tap / 3, kw = tap - 3 * kh;
            if constexpr (PIL && !LATE) compute_pil(tap % S, buf, kh, kw, tap == 8 ? buf ^ 1 : buf, issue_dma);
            else {
                issue_dma();
                if constexpr (LATE) compute_late(tap % S, buf, TWO_D ? kh * PW + kw : kh * p.qW + kw);
                else compute(tap % S, buf, TWO_D ? kh * PW + kw : kh * p.qW + kw);
            }
        }
    };
    if constexpr (PIL) {        // the position fragments of the very first tap: the patch is the oldest DMA in flight (W_FIRST: the youngest)
        if (has_a && !W_FIRST) tap_wait_vm<2 * JA>(); else tap_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int b = 0; b < TP; ++b) G[b] = *reinterpret_cast<const uint4v *>(g_ptr(0, wn * (TP * 16) + b));
        __builtin_amdgcn_sched_barrier(0);
    }
    auto run_all = [&](auto latec) {
        if constexpr (STAG) init_acc();
        if constexpr (decltype(latec)::value) {
#pragma unroll
            for (int a = 0; a < TMH; ++a) ha[a] = uint4v{0u, 0u, 0u, 0u};       // (the first tap has no predecessor: its deferred half adds 0 x 0)
#pragma unroll
            for (int b = 0; b < TP; ++b) hb[b] = uint4v{0u, 0u, 0u, 0u};
        }
        for (int c = c_begin; c < C; c += 2) {
            if constexpr (S2) {
                run_slice_s2(c, std::integral_constant<int, 0>(), latec);
                if (c + 1 < C) run_slice_s2(c + 1, std::integral_constant<int, 1>(), latec);
            } else {
                run_slice(c, std::integral_constant<int, 0>(), latec);
                if (c + 1 < C) run_slice(c + 1, std::integral_constant<int, 1>(), latec);
            }
            if constexpr (F32) flush_acc<TM, TP>(acc, acc2);        // two 16-channel slices x 9 taps = 288 k per chain
        }
    };
    if constexpr (STAG) {
        if (late) {
            run_all(std::true_type());
            late_finish();              // the second half of the last tap
        } else {
            run_all(std::false_type());
        }
    } else {
        if constexpr (PAIR) { if (pair_half == 1) __builtin_amdgcn_s_barrier(); }      // half 1 runs one barrier = half a tap behind half 0 ...
        run_all(std::false_type());
        if constexpr (PAIR) { if (pair_half == 0) __builtin_amdgcn_s_barrier(); }      // ... which therefore owes one at the end
    }
    if constexpr (!F32) tap_mfma_drain<TM, TP>(acc);
    if constexpr (F32) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TP; ++b) acc[a][b] = acc2[a][b];
    }
    (void)KT;
#ifdef YOLO_EXPERIMENT
    const unsigned long long t_loop = p.trace ? wall_clock64() : 0ull;
#endif
    if constexpr (SPLITK) {
        if constexpr (TP == 2) {            // (the 128 x 256 instantiation exists for the in-launch pair only)
            if (!p.pair) {
                conv_store_partial<TM, TP, PADQ, PIL>(p, acc, n0 + wm * (TM * 16) + fq * CH, q0 + wn * (TP * 16), fr, block_y);
                return;
            }
        }
        // ---- split-K inside the launch (two halves per tile) -----------------------------------------------------------------
        // Hand-off per cdna_hip_programming.md Guideline 16 / "In-launch split-K reduction": every wave stores its accumulators
        // WRITE-THROUGH (sc1: no release fence), drains its own stores, the workgroup meets at a barrier, ONE lane takes the
        // ticket (relaxed agent-scope atomic); the second arriver acquires (one agent-scope fence by that lane, drained before
        // the barrier that releases the other waves) and reads the first arriver's slab with sc1 loads.  Placement-independent;
        // fp32 addition commutes, so the result does not depend on which half arrives last.  The counter returns to 0.
        // More than two splits (round 4: the split-K launches of the small maps at batch 1-4 without their reduce launch): the LAST
        // arriver sums the slabs of ALL splits in split order -- its own included, from memory: fp32 addition is not associative, and the
        // result must not depend on who arrives last.
        constexpr uint32_t SLAB = (uint32_t)NA * NB * 4;
        const __amdgpu_buffer_rsrc_t rs_part = __builtin_amdgcn_make_buffer_rsrc(p.part, 0, p.part_bytes, 0x00020000);
        const uint32_t nsplit = (uint32_t)nsplit_y;
        const uint32_t mine = ((uint32_t)raw_x * nsplit + (uint32_t)block_y) * SLAB, other = ((uint32_t)raw_x * nsplit + (1u - (uint32_t)block_y)) * SLAB;
        typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u4;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TP; ++b)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, acc[a][b]), rs_part,
                                                       mine + (uint32_t)((((wave * TM + a) * TP + b) * 64 + lane) * 16), 0, 16 /* sc1 */);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave
        __syncthreads();                                        // ... and nobody reads the LDS rings any more
        int *const flag = reinterpret_cast<int *>(smem);
        if (tid == 0) {
            int *const ticket = p.pair_cnt + (size_t)raw_x * kCandCountStride;      // (a 128-byte line per tile: the tickets of a launch
            // arrive within a microsecond of each other, and 32 of them in one line queue behind one L2 channel)
            const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == (int)nsplit - 1) {
                __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            *flag = t;
        }
        __syncthreads();
        if (*flag != (int)nsplit - 1) return;       // not the last arriver: its share is published
        if (nsplit > 2) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TP; ++b) acc[a][b] = float4v{0.f, 0.f, 0.f, 0.f};
            for (uint32_t sp = 0; sp < nsplit; ++sp) {
                const uint32_t base = ((uint32_t)raw_x * nsplit + sp) * SLAB;
#pragma unroll
                for (int a = 0; a < TM; ++a) {
                    u4 v[TP];
#pragma unroll
                    for (int b = 0; b < TP; ++b)
                        v[b] = __builtin_amdgcn_raw_buffer_load_b128(rs_part, base + (uint32_t)((((wave * TM + a) * TP + b) * 64 + lane) * 16), 0, 16 /* sc1 */);
#pragma unroll
                    for (int b = 0; b < TP; ++b) acc[a][b] += __builtin_bit_cast(float4v, v[b]);
                }
            }
            conv_epilogue<T, TM, TP, PADQ, true, PIL>(p, acc, n0 + wm * (TM * 16) + fq * CH, q0 + wn * (TP * 16), fr);
            return;
        }
#pragma unroll
        for (int a = 0; a < TM; ++a) {      // TP loads in flight at a time (all TM * TP at once would need 64 more registers)
            u4 v[TP];
#pragma unroll
            for (int b = 0; b < TP; ++b)
                v[b] = __builtin_amdgcn_raw_buffer_load_b128(rs_part, other + (uint32_t)((((wave * TM + a) * TP + b) * 64 + lane) * 16), 0, 16 /* sc1 */);
#pragma unroll
            for (int b = 0; b < TP; ++b) acc[a][b] += __builtin_bit_cast(float4v, v[b]);
            __builtin_amdgcn_sched_barrier(0);
        }
        conv_epilogue<T, TM, TP, PADQ, true, PIL>(p, acc, n0 + wm * (TM * 16) + fq * CH, q0 + wn * (TP * 16), fr);
        return;
    } else {
        // (a template mode, not a run-time branch: with the branch in the code the 128 x (16 x 16) tile spilled 28 VGPRs)
        if constexpr (MODE == 3) conv_epilogue_pool2<T, TM, TP>(p, acc, n0 + wm * (TM * 16) + fq * CH, q0 + wn * (TP * 16), fr);
        else if constexpr (FUSE2) {
            static_assert(!FUSE2 || 2 * LDS_BYTES <= 163840, "two workgroups per CU");
            if constexpr (S2) conv_epilogue_fused_1x1<PADQ, false>(p, acc, q0, wm, wn, wave, lane, smem);    // (the stride-2 conv into a stage: no residual)
            else conv_epilogue_fused_1x1<PADQ, true>(p, acc, q0, wm, wn, wave, lane, smem);       // (the residual block's 3x3: with residual)
        }
        else if constexpr (FAST) { if (!PAIR || tile_valid) conv_epilogue_fast<TM, TP, PADQ, PIL, LDAUX>(p, acc, n0 + wm * (TM * 16) + fq * CH, q0 + wn * (TP * 16), fr); }
        else conv_epilogue<T, TM, TP, PADQ, true, PIL>(p, acc, n0 + wm * (TM * 16) + fq * CH, q0 + wn * (TP * 16), fr);
    }
#ifdef YOLO_EXPERIMENT
    if (p.trace && tid == 0) {          // YOLO_CONV_TRACE: phase timestamps (100 MHz) + placement of wave 0 of every block
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *r = p.trace + (size_t)raw_x * 8;
        r[0] = t_start; r[1] = t_setup; r[2] = t_loop; r[3] = wall_clock64();
        r[4] = __builtin_amdgcn_s_getreg(0xF804);      // HW_ID
        r[5] = __builtin_amdgcn_s_getreg(0xF814);      // XCC_ID
        r[6] = (unsigned long long)bid;
        r[7] = (unsigned long long)clock64() - c_start;     // shader-clock cycles of the block (vs r[3] - r[0] at 100 MHz)
    }
#endif
}

}  // namespace yolo
