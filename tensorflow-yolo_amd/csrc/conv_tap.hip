// 3x3 / stride-1 implicit-GEMM conv (fp16) with TAP REUSE of the pixel operand.
//
// Why: the LDS-DMA kernels of conv_dma.hip fetch the pixel tile once PER TAP (im2col on the fly), and their K
// loop is bound by the L2 -> LDS path: 20 TB/s chip-wide = ~78 GB/s per CU measured with the MFMAs switched
// off (profiles/r01_ablation.md; the guide's L2-served gather-into-LDS figure is 66-73 GB/s per CU), against
// ~85-130 flop/byte of those tiles.  Here the input patch of a block is loaded ONCE per 32-channel slice and
// all nine taps read their MFMA B operand from it at a row offset, so a 128-cout x 256-pixel block moves
// 8 KiB of weights per tap + ~26 KiB of pixels per nine taps: ~190 flop/byte, 2.2x fewer L2 -> LDS bytes.
//
// Pixel indexing: positions q of a PADDED-LINEAR grid [n][y in 0..H][x in 0..W] -- one shared pad column after
// every image row and one shared pad row after every image.  Tap (kh, kw) of position q is position
// q + (kh-1)(W+1) + (kw-1), for every q, so a block owning NB consecutive positions keeps the NB + 2W + 4
// positions around them in LDS (64-byte rows = 32 channels) and tap t reads rows shifted by kh (W+1) + kw.
// Pad positions are filled with zeros by the LDS-DMA range check (per-lane source offset = invalid) -- this is
// the conv's zero padding -- and as OUTPUT positions they are computed and dropped in the epilogue
// ((H+1)(W+1)/(HW) - 1 wasted MFMA work: 2.6 % at 76x76, 10.8 % at 19x19).
//
// LDS swizzle of the patch: logical 16-byte chunk c of row R sits at chunk c ^ (2 * ((R >> 2) & 1)).  Unlike
// the table swizzle of the weight tile this one stays conflict-free for ds_read_b128 under ANY row shift: the
// hardware services lanes {0-3, 12-15, 20-27} (k-chunk q for fragment rows 0-3, 12-15, chunk q^1 for rows
// 4-11) together, the four rows of one residue class mod 4 are consecutive values of R >> 2, and
// {s(u), s(u+1)^1, s(u+2)^1, s(u+3)} is a permutation of 0..3 for s(u) = 2 (u & 1) and every u.
//
// K loop: for each 32-channel slice c: for tap 0..8 (unrolled): weights of (tap, c) come through a 3-slot
// LDS-DMA ring (slot = tap % 3), the patch of slice c+1 is fetched into the other patch buffer during the taps
// of slice c.  One barrier per tap; vmcnt waits are counted so that the patch may stay in flight for three taps.
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
#ifndef YOLO_TAP_STAGGER
#define YOLO_TAP_STAGGER 1
#endif

namespace yolo {

namespace {

typedef __attribute__((address_space(3))) void tap_lds_void;

// voff: per-lane byte offset (loop invariant, range-checked: an invalid offset writes zeros); soff: wave-uniform
// byte offset of the K position, added by the hardware after the range check -> no per-tap address registers.
__device__ __forceinline__ void tap_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char *lds_dst, uint32_t voff, uint32_t soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (tap_lds_void *)lds_dst, 16, voff, soff, 0, 0);
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
template <bool F32, int WM, int WN, int TM, int TP, int PRG, int OCC, int MODE, bool SPLITK, bool FAST = false, bool FUSE2 = false>
__global__ void __launch_bounds__(WM * WN * 64, OCC) conv3x3_tap_kernel(const ConvParams p) {
    typedef typename std::conditional<F32, float, _Float16>::type T;
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
    constexpr int LDS_BYTES = FUSE2 && kFuse2LdsBytes > S * A_BYTES + 2 * P_BYTES ? kFuse2LdsBytes : S * A_BYTES + 2 * P_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS_BYTES];
    unsigned char *const smemP = smem + S * A_BYTES;

    const int tid = threadIdx.x;
#ifdef YOLO_EXPERIMENT      // block trace (tools/trace_blocks.py); not in the product build
    const unsigned long long t_start = p.trace ? wall_clock64() : 0ull;
    const unsigned long long c_start = p.trace ? (unsigned long long)clock64() : 0ull;
    // experiment: (dbg & 64) the workgroups with an odd threadgroup id run at wave priority 3 -- does priority decide who gets the CU?
    if ((p.dbg & 64) && ((__builtin_amdgcn_s_getreg(0xF804) >> 16) & 1u)) __builtin_amdgcn_s_setprio(3);
    // experiment (tools/offset_probe.py): the workgroups with an odd threadgroup id on their CU start (dbg >> 8) x 0.25 us late
    if ((p.dbg >> 8) > 0 && ((__builtin_amdgcn_s_getreg(0xF804) >> 16) & 1u)) {
        const unsigned long long until = wall_clock64() + (unsigned long long)(p.dbg >> 8) * 25ull;
        while (wall_clock64() < until) __builtin_amdgcn_s_sleep(4);
    }
#endif
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int bid = xcd_remap(blockIdx.x, p.n_blocks);
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
    const int c_begin = SPLITK ? (int)blockIdx.y * p.kunits : 0;
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
            if (j + 1 < JP || jp_full) tap_dma16(rs_in, smemP + buf * P_BYTES + (j * NW + wave) * 1024, b_off[j], koff);
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
            if ((TP != 2 || p.pair) && blockIdx.y == 0) {    // in-launch pair: the bias rides in half 0
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
#pragma unroll
                for (int a = a0; a < a0 + AH; ++a)
#pragma unroll
                    for (int b = 0; b < TP; ++b) tap_mfma<T>(acc[a][b], fa[a], G[(b + kw) % TP]);
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
        run_all(std::false_type());
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
                conv_store_partial<TM, TP, PADQ, PIL>(p, acc, n0 + wm * (TM * 16) + fq * CH, q0 + wn * (TP * 16), fr, (int)blockIdx.y);
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
        const uint32_t nsplit = gridDim.y;
        const uint32_t mine = ((uint32_t)blockIdx.x * nsplit + blockIdx.y) * SLAB, other = ((uint32_t)blockIdx.x * nsplit + (1u - blockIdx.y)) * SLAB;
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
            int *const ticket = p.pair_cnt + (size_t)blockIdx.x * kCandCountStride;      // (a 128-byte line per tile: the tickets of a launch
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
                const uint32_t base = ((uint32_t)blockIdx.x * nsplit + sp) * SLAB;
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
        else if constexpr (FAST) conv_epilogue_fast<TM, TP, PADQ, PIL>(p, acc, n0 + wm * (TM * 16) + fq * CH, q0 + wn * (TP * 16), fr);
        else conv_epilogue<T, TM, TP, PADQ, true, PIL>(p, acc, n0 + wm * (TM * 16) + fq * CH, q0 + wn * (TP * 16), fr);
    }
#ifdef YOLO_EXPERIMENT
    if (p.trace && tid == 0) {          // YOLO_CONV_TRACE: phase timestamps (100 MHz) + placement of wave 0 of every block
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long *r = p.trace + (size_t)blockIdx.x * 8;
        r[0] = t_start; r[1] = t_setup; r[2] = t_loop; r[3] = wall_clock64();
        r[4] = __builtin_amdgcn_s_getreg(0xF804);      // HW_ID
        r[5] = __builtin_amdgcn_s_getreg(0xF814);      // XCC_ID
        r[6] = (unsigned long long)bid;
        r[7] = (unsigned long long)clock64() - c_start;     // shader-clock cycles of the block (vs r[3] - r[0] at 100 MHz)
    }
#endif
}

// ---- persistent ("stream") form ---------------------------------------------------------------------------------------------
// Block traces (tools/trace_blocks.py, profiles/r03_ablation.md) show what a workgroup of the kernel above does outside its K loop:
// ~0.8 us of setup, ~1.5-2 us until its first patch + weights have landed, 3.6-3.9 us of epilogue -- 20 % of its life at 76 x 76,
// half of it at 304 x 304 (K = 288) -- and a CU is only at full rate while BOTH of its workgroups are inside their K loops.
// Here the 512 (or 256) resident workgroups stay and walk tiles blockIdx, blockIdx + grid, ...: the (tile, channel slice) items
// form ONE stream through the same LDS rings, so the patch and the first two weight tiles of the next tile are requested during the
// last slice of the current one and are in LDS when its epilogue ends; the geometry of the next tile is computed under the K loop.
// The epilogue's loads and stores go through buffer instructions that every wave issues unconditionally (invalid lanes carry an
// out-of-range offset), so their number is a compile-time constant and the counted `s_waitcnt vmcnt` of the two taps that follow an
// epilogue can leave the stores in flight.  The bias of every cout lives in LDS (the accumulators restart from it without a VMEM
// load, which would otherwise have to wait for those stores).  fp16, whole K, OUT_NORMAL, 16-byte aligned views only.
template <int WM, int WN, int TM, int TP, int PRG, int OCC, int MODE>
__global__ void __launch_bounds__(512, OCC) conv3x3_tap_stream_kernel(const ConvParams p) {
    typedef _Float16 T;
    constexpr int NW = 8, S = 3, ROWB = 64;
    constexpr int NA = WM * TM * 16;
    constexpr int NB = WN * TP * 16;
    constexpr int JA_TOT = NA / 16;
    constexpr int JA = (JA_TOT + NW - 1) / NW;
    constexpr int PW = 24;
    constexpr int FROW = MODE == 2 ? PW : 16;
    constexpr int JP = (PRG + NW - 1) / NW;
    constexpr int CH = 4 * TM;
    constexpr int EPC = 8;
    constexpr int NQ = CH / EPC;                // 16-byte chunks of a lane's couts
    constexpr int NST = TP * NQ;                // store instructions of one wave's epilogue (always issued)
    constexpr int A_BYTES = NA * ROWB;
    constexpr int P_BYTES = PRG * 1024;
    constexpr int BIAS_FLOATS = kTapStreamBiasFloats;
    static_assert(WM * WN == NW && (JA_TOT % NW == 0 || JA_TOT < NW), "tile / wave mapping");
    __shared__ __attribute__((aligned(16))) unsigned char smem[S * A_BYTES + 2 * P_BYTES + BIAS_FLOATS * 4];
    unsigned char *const smemP = smem + S * A_BYTES;
    float *const sbias = reinterpret_cast<float *>(smem + S * A_BYTES + 2 * P_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const bool has_a = JA_TOT % NW == 0 || wave < JA_TOT;   // wave-uniform
    const int fr = lane & 15, fq = lane >> 4;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wgt), 0, p.wgt_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.res ? p.res : p.out), 0, p.res ? p.res_bytes : 0u, 0x00020000);

    for (int i = tid; i < p.cout_pad && i < BIAS_FLOATS; i += 512) sbias[i] = p.bias[i];      // (read after the first tap's barrier at the earliest)

    // ---- lane constants of the DMA geometry --------------------------------------------------------
    const int lrow = lane >> 2;
    uint32_t a_off[JA];         // without the cout tile: n0 * wrow_bytes rides in the scalar offset of the DMA
#pragma unroll
    for (int j = 0; j < JA; ++j) {
        const int r = (j * NW + wave) * 16 + lrow;
        const int ws = r / (TM * 16), R = r % (TM * 16);
        const int tm = R >> 4, g4 = (R >> 2) & 3, jj = R & 3;
        const int ch = ws * (TM * 16) + g4 * CH + 4 * tm + jj;
        a_off[j] = (uint32_t)ch * p.wrow_bytes + (uint32_t)(((lane & 3) ^ tap_swz_w(lrow)) << 4);
    }
    constexpr int JP_FULL = PRG - (JP - 1) * NW;
    const bool jp_full = wave < JP_FULL;
    const uint32_t csw_p = (uint32_t)(((lane & 3) ^ (((lrow >> 2) & 1) << 1)) << 4);
    uint32_t b_off[JP];
    // per-tile: patch offsets of tile `bid` (remapped linear tile id) -> b_off; returns q0 and n0 of the tile
    auto geometry = [&](int bid, int &q0, int &n0) __attribute__((always_inline)) {
        const int mt = (int)fdiv((uint32_t)bid, p.dtiles_n);
        n0 = (bid - mt * p.n_tiles_n) * NA;
        q0 = mt * NB;
        int t2_n = 0, t2_y0 = 0, t2_x0 = 0;
        if (MODE == 2) {
            t2_n = (int)fdiv((uint32_t)mt, p.dqHW);
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
            if (MODE == 2) {
                const int R = g * 16 + lrow;
                const int pr = R / PW, pc = R - pr * PW;
                n = t2_n; y = t2_y0 - 1 + pr; x = t2_x0 - 1 + pc;
                ok = g < PRG && pc < 18 && pr < NB / 16 + 2 && t2_n * p.HoWo < p.M && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            } else {
                const int q = q0 - (p.qW + 1) + g * 16 + lrow;
                ok = g < PRG && q >= 0 && q < p.Mq;
                const int qq = ok ? q : 0;
                n = (int)fdiv((uint32_t)qq, p.dqHW);
                const int r = qq - n * p.qHW;
                y = (int)fdiv((uint32_t)r, p.dqW);
                x = r - y * p.qW;
                ok = ok && x < p.W && y < p.H;
            }
            const long long e = (long long)n * p.in_img_stride + ((long long)y * p.W + x) * p.in_ld + p.in_coff;
            b_off[j] = ok ? (uint32_t)(e * 2) + csw_p : YOLO_INVALID_OFF;
        }
    };

    const int C = p.cin_chunks >> 2;            // channel slices per tile
    auto issue_patch = [&](int c, int buf) __attribute__((always_inline)) {
        const uint32_t koff = (uint32_t)c * ROWB;
#pragma unroll
        for (int j = 0; j < JP; ++j)
            if (j + 1 < JP || jp_full) tap_dma16(rs_in, smemP + buf * P_BYTES + (j * NW + wave) * 1024, b_off[j], koff);
    };
    auto issue_weights = [&](int tap, int c, int n0, int slot) __attribute__((always_inline)) {
        const uint32_t ka = (uint32_t)(tap * p.cin_chunks + 4 * c) * 16 + (uint32_t)n0 * p.wrow_bytes;
        if (has_a) {
#pragma unroll
            for (int j = 0; j < JA; ++j) tap_dma16(rs_w, smem + slot * A_BYTES + (j * NW + wave) * 1024, a_off[j], ka);
        }
    };

    float4v acc[TM][TP];
    const int a_frag = (wm * TM * 16 + fr) * ROWB + (((fq ^ tap_swz_w(fr)) & 3) << 4);
    const int rb = wn * TP * FROW + fr;
    const int c_lane = wm * (TM * 16) + fq * CH;        // first cout of this lane inside the cout tile
    auto compute = [&](int slot, int buf, int shift) __attribute__((always_inline)) {
        const unsigned char *A = smem + slot * A_BYTES + a_frag;
        const int R = rb + shift;
        const unsigned char *B = smemP + buf * P_BYTES + (R << 6) + ((fq << 4) ^ ((R & 4) << 3));
        uint4v fa[TM], fb[TP];
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[a] = *reinterpret_cast<const uint4v *>(A + a * 16 * ROWB);
#pragma unroll
        for (int b = 0; b < TP; ++b) fb[b] = *reinterpret_cast<const uint4v *>(B + b * FROW * ROWB);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TP; ++b) tap_mfma<T>(acc[a][b], fa[a], fb[b]);
    };
    // epilogue of one tile: leaky, + residual, fp16, 16-byte stores; NQ * TP loads (if any) and NST stores per wave, always
    auto epilogue = [&](int q0, int n0) __attribute__((always_inline)) {
        // the residual of DEPTH fragments is in flight at a time: all of them for the TP = 2 tile; two for the TP = 4 tiles, whose 128
        // registers cannot hold 32 residual registers next to the K loop's state that has to survive the epilogue here
        constexpr int DEPTH = TP > 2 ? 2 : TP;
        const int cbase = n0 + c_lane;
        const bool c_ok = cbase < p.Cout;
        uint32_t ooff[DEPTH];
        uint4v rv[DEPTH][NQ];
        auto request = [&](int b, int slot) __attribute__((always_inline)) {
            int n, rem, oy, ox;
            const bool ok = conv_decode_pixel<MODE>(p, q0 + wn * (TP * 16) + b * 16 + fr, n, rem, oy, ox) && c_ok;
            const long long o = ((long long)n * p.out_img_stride + (long long)rem * p.out_ld + cbase) * 2;
            ooff[slot] = ok ? (uint32_t)o : YOLO_INVALID_OFF;
            if (p.has_res) {
                const long long ro = ((long long)n * p.res_img_stride + (long long)rem * p.res_ld + cbase) * 2;
                const uint32_t roff = ok ? (uint32_t)ro : YOLO_INVALID_OFF;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    rv[slot][q] = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(rs_res, roff, q * 16, 0));
            }
        };
#pragma unroll
        for (int b = 0; b < DEPTH; ++b) request(b, b);
#pragma unroll
        for (int b = 0; b < TP; ++b) {
            const int slot = b % DEPTH;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                T t[EPC], r[EPC];
                if (p.has_res) __builtin_memcpy(r, &rv[slot][q], 16);
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const int i = q * EPC + e;
                    float x = acc[i >> 2][b][i & 3];
                    x = p.leaky ? fmaxf(0.1f * x, x) : x;
                    if (p.has_res) x += (float)r[e];
                    t[e] = (T)x;
                }
                uint4v u;
                __builtin_memcpy(&u, t, 16);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, u), rs_out, ooff[slot], q * 16, 0);
            }
            if (b + DEPTH < TP) {
                __builtin_amdgcn_sched_barrier(0);      // (keeps the next request behind this fragment's use of the slot)
                request(b + DEPTH, slot);
            }
        }
    };
    auto init_acc = [&](int n0, bool from_lds) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const float4v bv = from_lds ? *reinterpret_cast<const float4v *>(sbias + n0 + c_lane + 4 * a)
                                        : *reinterpret_cast<const float4v *>(p.bias + n0 + c_lane + 4 * a);
#pragma unroll
            for (int b = 0; b < TP; ++b) acc[a][b] = bv;
        }
    };

    // ---- the stream ------------------------------------------------------------------------------------
    int it = blockIdx.x;                       // this workgroup's tiles: it, it + grid, ... (< n_blocks)
    int q0_cur, n0_cur, q0_nxt = 0, n0_nxt = 0;
    geometry(xcd_remap(it, p.n_blocks), q0_cur, n0_cur);
    init_acc(n0_cur, false);
    issue_patch(0, 0);
    issue_weights(0, 0, n0_cur, 0);
    issue_weights(1, 0, n0_cur, 1);
    int c = 0;
    bool after_epi = false;                    // the previous item ended with an epilogue: NST stores sit between its DMAs and ours
    bool done = false;

    // one (tile, slice) item; the patch buffer index is a compile-time constant
    auto run_item = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int buf = decltype(bufc)::value;
        const bool last_c = c + 1 == C;
        const int it_n = last_c ? it + (int)gridDim.x : it;
        const bool more = !last_c || it_n < p.n_blocks;
        const int c_n = last_c ? 0 : c + 1;
        if (!last_c) { n0_nxt = n0_cur; q0_nxt = q0_cur; }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const bool last = !more && tap == 8;
            const bool with_patch = more && (tap == 1 || tap == 2);
            const bool st = after_epi && tap < 2;      // the epilogue's stores are younger than the DMAs this tap waits for
            if (last) tap_wait_vm<0>();
            else if (has_a) {
                if (!with_patch) { if (st) tap_wait_vm<JA + NST>(); else tap_wait_vm<JA>(); }
                else if (jp_full) { if (st) tap_wait_vm<JA + JP + NST>(); else tap_wait_vm<JA + JP>(); }
                else { if (st) tap_wait_vm<JA + JP - 1 + NST>(); else tap_wait_vm<JA + JP - 1>(); }
            } else {
                if (!with_patch) { if (st) tap_wait_vm<NST>(); else tap_wait_vm<0>(); }
                else if (jp_full) { if (st) tap_wait_vm<JP + NST>(); else tap_wait_vm<JP>(); }
                else { if (st) tap_wait_vm<JP - 1 + NST>(); else tap_wait_vm<JP - 1>(); }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            if (tap == 0 && more) {
                if (last_c) geometry(xcd_remap(it_n, p.n_blocks), q0_nxt, n0_nxt);      // b_off of the finished tile is dead: its last patch is in LDS
            }
            {   // weights two taps ahead (the stream continues into the next item)
                if (tap + 2 < 9) issue_weights(tap + 2, c, n0_cur, (tap + 2) % S);
                else if (more) issue_weights(tap + 2 - 9, c_n, n0_nxt, (tap + 2) % S);
            }
            if (tap == 0 && more) issue_patch(c_n, buf ^ 1);
            const int kh = tap / 3, kw = tap - 3 * kh;
            compute(tap % S, buf, MODE == 2 ? kh * PW + kw : kh * p.qW + kw);
        }
        after_epi = false;
        if (last_c) {
            tap_mfma_drain<TM, TP>(acc);
            epilogue(q0_cur, n0_cur);
            if (!more) { done = true; return; }
            it = it_n; q0_cur = q0_nxt; n0_cur = n0_nxt;
            init_acc(n0_cur, true);
            after_epi = true;
        }
        c = c_n;
    };
    for (;;) {
        run_item(std::integral_constant<int, 0>());
        if (done) break;
        run_item(std::integral_constant<int, 1>());
        if (done) break;
    }
}

// (6 = 256 couts x 224 positions, one workgroup per CU: 19 x 19 maps at batch 32 are 12 800 padded positions -> 58 x 4 = 232
// tiles on 256 CUs, where the 256-position tiles leave 200 or 400 workgroups on 256 / 512 slots)
// variants: 0 = 128 couts x 256 positions, 1 = 256 x 256 (one workgroup per CU), 2 = 128 x 192, 3 = 128 x 128 (smaller
// position tiles fill the 512 workgroup slots of the chip better on small feature maps), all padded-linear;
// 4 = 128 x (16 x 16) and 5 = 64 x (16 x 16) 2-D tiles for maps wider than 78 (any width)
// 7 = 128 couts x (8 x 16) 2-D tile at THREE workgroups per CU (48 KiB LDS, <= 80 VGPRs) for wide maps with a short K, where
// a workgroup spends as long in setup + epilogue as in its K loop (152 x 152 64 -> 128: block trace in profiles/r03_ablation.md)
// 8 = 32 couts x (16 x 16): the one 3x3 layer with 32 filters behind the first conv (tiny-YOLOv2 16 -> 32 at 208 x 208; float32 MFMA is 1/16 of
// fp16's, so the 64-cout tile's idle half would double a launch that is MFMA-bound)
// 9 = 128 couts x 384 positions, IMAGE-ALIGNED, one workgroup per CU: a tile is one whole image of a map with H (W+1) <= 384 (19 x 19:
// 380), tile m starts at position m (H+1)(W+1) -- the shared pad row behind every image is never computed (6 % of the positions are
// padding instead of 10.8 %) and 19 x 19 at batch 32 is 32 x 8 = 256 tiles: every CU busy, 14 % less work per CU than the 232 tiles
// of variant 6
// 10 = 128 x 256 and 11 = 128 x 384 image-aligned for 3x3 / STRIDE 2 (MODE 4: parity planes of the input, see the kernel): fp16 only
// 12 = 128 x 192 image-aligned: one 12 x 12 or 13 x 13 image per tile (YOLOv2-416 / YOLOv3-416 tails: 13 x 14 = 182 of 192 positions
// real, where 256-position tiles of the padded-linear grid compute 23 % padding); with the in-launch pair split 16 images x 8 cout
// tiles x 2 K halves = 256 workgroups
// 13 = variant 10 with a patch of 26 row groups (output maps up to 158 wide): the stride-2 conv into the 152 x 152 stage, whose
// workgroups hold all 128 couts of 256 positions -- the one stride-2 tile with the back-to-back 1x1 instantiation (FUSE2)
// (round 5, measured and NOT kept -- profiles/r05_ablation.md, source in commit 57e38ee: variant 0's block tile as four fat waves of 64 couts x 128
// positions with the position fragments prefetched across the barrier, +0.5 ... +2.5 % slower; variant 9's 128 x 384 tile on the padded-linear grid
// for 38 x 38 maps -- 508 tiles on 256 CUs -- +4.3 % slower than the 764 workgroups of variant 0; position fragments of the taps kw = 1, 2 by DPP lane
// shifts instead of LDS reads, correct and +5 ... +9 % slower)
static const int kTapNB[] = {256, 256, 192, 128, 256, 256, 224, 128, 256, 384, 256, 384, 192, 256};
static const int kTapPRG[] = {26, 26, 26, 28, 27, 27, 17, 12, 27, 27, 21, 26, 14, 26};
static const int kTapTP[] = {4, 4, 3, 2, 4, 2, 7, 2, 2, 6, 4, 6, 3, 4};        // TP of the variant (YOLO_TAP_VARIANTS below): position fragments per wave
static const int kTapVariants = 14;
static const bool kTapF32[] = {false, false, false, true, false, true, false, false, true, false, false, false, false, false};      // float32 tiles: TP <= 2 (second-level accumulator)
bool conv_tap_image_aligned(int variant) { return variant == 9 || variant == 11 || variant == 12; }
bool conv_tap_stride2(int variant) { return variant == 10 || variant == 11 || variant == 13; }
bool conv_tap_splitk_ok(int variant) { return variant == 3; }      // the 128 x 128 tile has the (two-pass) split-K instantiation
bool conv_tap_pair_ok(int variant, bool f32) { return variant == 3 || ((variant == 0 || variant == 12) && !f32); }   // in-launch pair split: also the fp16 128 x 256 and image-aligned 128 x 192 tiles
bool conv_tap_is2d(int variant) { return variant == 4 || variant == 5 || variant == 7 || variant == 8; }
bool conv_tap_f32_ok(int variant) { return variant >= 0 && variant < kTapVariants && kTapF32[variant]; }
bool conv_tap_fits(int variant, int W) {
    if (variant < 0 || variant >= kTapVariants) return false;
    if (conv_tap_is2d(variant)) return true;
    if (conv_tap_stride2(variant)) {        // W = the INPUT's width (even); the position grid is the output's
        if (W & 1) return false;
        W >>= 1;
    }
    // (the image-aligned tile: square maps of 17 .. 19 -- a whole image per tile with at least 80 % of the positions real)
    if (conv_tap_image_aligned(variant) && (W * (W + 1) > kTapNB[variant] || W * (W + 1) * 5 < kTapNB[variant] * 4)) return false;
    if (conv_tap_stride2(variant)) return kTapNB[variant] + W + 2 <= kTapPRG[variant] * 16;
    // (position-interleaved fragments: the patch buffer holds TP planes of PRG * 16 / TP whole rows)
    const int tp = kTapTP[variant], rows = YOLO_TAP_PIL ? kTapPRG[variant] * 16 / tp * tp : kTapPRG[variant] * 16;
    return kTapNB[variant] + 2 * W + 4 <= rows;
}

// variant id, then the template arguments after F32: WM, WN, TM, TP, PRG, OCC, MODE (written with ", " so that the
// stringified list equals the demangled symbol)
#define YOLO_TAP_VARIANTS(X) \
    X(0, 2, 4, 4, 4, 26, 4, 1) \
    X(1, 2, 4, 8, 4, 26, 2, 1) \
    X(2, 2, 4, 4, 3, 26, 4, 1) \
    X(3, 2, 4, 4, 2, 28, 4, 1) \
    X(4, 2, 4, 4, 4, 27, 4, 2) \
    X(5, 1, 8, 4, 2, 27, 4, 2) \
    X(6, 4, 2, 4, 7, 17, 2, 1) \
    X(7, 2, 4, 4, 2, 12, 6, 2) \
    X(8, 1, 8, 2, 2, 27, 4, 2) \
    X(9, 2, 4, 4, 6, 27, 2, 1) \
    X(10, 2, 4, 4, 4, 21, 4, 4) \
    X(11, 2, 4, 4, 6, 26, 2, 4) \
    X(12, 2, 4, 4, 3, 14, 4, 1) \
    X(13, 2, 4, 4, 4, 26, 4, 4)

const char *conv_tap_symbol(int variant, bool f32, bool fast) {
    switch (variant) {
#define X(id, ...) case id: return f32 ? "void yolo::conv3x3_tap_kernel<true, " #__VA_ARGS__ ", false, false, false>(yolo::ConvParams)" \
                                       : fast ? "void yolo::conv3x3_tap_kernel<false, " #__VA_ARGS__ ", false, true, false>(yolo::ConvParams)" \
                                              : "void yolo::conv3x3_tap_kernel<false, " #__VA_ARGS__ ", false, false, false>(yolo::ConvParams)";
        YOLO_TAP_VARIANTS(X)
#undef X
    default: return "";
    }
}

// the persistent form (conv3x3_tap_stream_kernel) is instantiated for variant 5 only (64 couts x 16 x 16 pixels, 93 VGPRs, no spill:
// 304 x 304 32 -> 64 at batch 32 0.242 -> 0.225 ms).  The TP = 4 tiles (variants 0 and 4) sit at the 128-register limit of two
// workgroups per CU: with the loop state of the stream they spill (19-23 VGPRs with all four residual fragments in flight, 5-17 with
// the two-deep residual pipeline the epilogue has for them), reloads land inside the K loop (each a `s_waitcnt vmcnt(0)` that drains
// the DMA pipeline) and the launches got SLOWER both times: 76 x 76 +15 % / +3 %, 152 x 152 +6.5 % / +13 % (profiles/r03_ablation.md).
// (round 5: with the MFMAs in place the 128 x 256 tile compiles in this form without a spill -- 123 registers -- and is still slower than the
// plain kernel: 76 x 76 +3 ... +5 %, 38 x 38 +-0, profiles/r05_ablation.md section 6; so it stays the 64-cout tile's alone)
#define YOLO_TAP_STREAM_VARIANTS(X) \
    X(5, 1, 8, 4, 2, 27, 4, 2)

const char *conv_tap_stream_symbol(int variant) {
    switch (variant) {
#define X(id, ...) case id: return "void yolo::conv3x3_tap_stream_kernel<" #__VA_ARGS__ ">(yolo::ConvParams)";
        YOLO_TAP_STREAM_VARIANTS(X)
#undef X
    default: return "";
    }
}

bool conv_tap_stream_ok(const ConvParams &p, int variant) {
    if (variant != 5) return false;
    if (p.f32 || p.out_f32 || p.ksplit > 1 || p.outmode != OUT_NORMAL || !p.vec_out || (p.has_res && (!p.vec_res || !p.res_bytes)) || !p.out_bytes) return false;
    if (p.Cout % 16 || (p.Cout + 127) / 128 * 128 > kTapStreamBiasFloats) return false;
    return true;
}

static hipError_t launch_conv_tap_stream(const ConvParams &p0, int variant, hipStream_t s) {
    ConvParams p = p0;
    p.cout_pad = (p.Cout + 127) / 128 * 128;
    const int slots = 512;              // two workgroups per CU (every stream variant is OCC 4)
    const dim3 grid((unsigned)(p.n_blocks < slots ? p.n_blocks : slots));
    switch (variant) {
#define X(id, ...) case id: hipLaunchKernelGGL((conv3x3_tap_stream_kernel<__VA_ARGS__>), grid, dim3(512), 0, s, p); break;
        YOLO_TAP_STREAM_VARIANTS(X)
#undef X
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_conv_tap(const ConvParams &p0, int variant, hipStream_t s) {
    ConvParams p = p0;
    static const bool no_fast_epi = getenv("YOLO_NO_FAST_EPI") != nullptr;        // A/B switch, read once (same results either way)
    // (variant 10, the two-per-CU stride-2 tile: its lean instantiation spills 13 registers at the 128-register limit; the generic one does not)
    p.fast_epi = (!no_fast_epi || p.fuse2) && conv_fast_epilogue_ok(p) && variant != 10 && (variant != 13 || p.fuse2) ? 1 : 0;      // (the fused pair exists in the lean form only)
    const bool s2 = conv_tap_stride2(variant);
    if (p.ksize != 3 || p.stride != (s2 ? 2 : 1) || p.pad != 1 || (p.cin_chunks & 3) || !conv_tap_fits(variant, p.W) || (p.f32 && !conv_tap_f32_ok(variant)))
        return hipErrorInvalidValue;
    if (s2 ? ((p.H & 1) || (p.W & 1) || p.Ho * 2 != p.H || p.Wo * 2 != p.W || p.f32 || p.ksplit > 1 || (p.fuse2 && variant != 13) || p.outmode == OUT_POOL2 || p.qW != p.Wo + 1)
           : (p.Ho != p.H || p.Wo != p.W))
        return hipErrorInvalidValue;
    if (conv_tap_image_aligned(variant) ? (p.q_stride != p.qHW || p.Ho * (p.Wo + 1) > kTapNB[variant]) : (!conv_tap_is2d(variant) && p.q_stride != kTapNB[variant]))
        return hipErrorInvalidValue;
    if (p.outmode == OUT_POOL2 && ((variant != 4 && variant != 5 && variant != 8) || (p.H & 1) || (p.W & 1) || p.has_res || p.ksplit > 1 || !p.vec_out || p.Cout % 16))
        return hipErrorInvalidValue;        // the fused pool lives in the 2-D tiles' epilogue only (plan.cpp asks for it accordingly)
    if (p.stream && conv_tap_stream_ok(p, variant)) return launch_conv_tap_stream(p, variant, s);
    const dim3 grid((unsigned)p.n_blocks, (unsigned)(p.ksplit > 1 ? p.ksplit : 1));
    if (p.ksplit > 1) {     // split-K instantiation (128 x 128 tile)
        if (!(p.pair ? conv_tap_pair_ok(variant, p.f32 != 0) : conv_tap_splitk_ok(variant)) || !p.part || p.kunits < 1 ||
            (long long)p.ksplit * p.kunits < (p.cin_chunks >> 2))
            return hipErrorInvalidValue;
        if (p.pair && (p.ksplit < 2 || (p.ksplit > 2 && variant != 3) || !p.pair_cnt || p.n_blocks > 512 ||        // (512 padded tickets: api.cpp kPairCounterBytes)
                       (unsigned long long)p.n_blocks * (unsigned long long)p.ksplit * 128ull * kTapNB[variant] * 4ull > p.part_bytes))
            return hipErrorInvalidValue;
        // (OCC 2 = up to 256 registers: the pair launches are <= 512 workgroups of half K on 256 CUs, and the 128 x 256 tile + the
        // hand-off state spills at the 128 registers of two-per-CU residency)
        if (variant == 0) hipLaunchKernelGGL((conv3x3_tap_kernel<false, 2, 4, 4, 4, 26, 2, 1, true>), grid, dim3(512), 0, s, p);
        else if (variant == 12) hipLaunchKernelGGL((conv3x3_tap_kernel<false, 2, 4, 4, 3, 14, 2, 1, true>), grid, dim3(512), 0, s, p);
        else if (p.f32) hipLaunchKernelGGL((conv3x3_tap_kernel<true, 2, 4, 4, 2, 28, 4, 1, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv3x3_tap_kernel<false, 2, 4, 4, 2, 28, 4, 1, true>), grid, dim3(512), 0, s, p);
        return hipGetLastError();
    }
    if (p.outmode == OUT_POOL2) {       // MODE 3 = the 2-D tile + the max-pool in the epilogue: variants 4 (fp16) and 5 (fp16, float32)
        if (variant == 4 && !p.f32) hipLaunchKernelGGL((conv3x3_tap_kernel<false, 2, 4, 4, 4, 27, 4, 3, false>), grid, dim3(512), 0, s, p);
        else if (variant == 5 && !p.f32) hipLaunchKernelGGL((conv3x3_tap_kernel<false, 1, 8, 4, 2, 27, 4, 3, false>), grid, dim3(512), 0, s, p);
        else if (variant == 5) hipLaunchKernelGGL((conv3x3_tap_kernel<true, 1, 8, 4, 2, 27, 4, 3, false>), grid, dim3(512), 0, s, p);
        else if (variant == 8 && !p.f32) hipLaunchKernelGGL((conv3x3_tap_kernel<false, 1, 8, 2, 2, 27, 4, 3, false>), grid, dim3(512), 0, s, p);
        else if (variant == 8) hipLaunchKernelGGL((conv3x3_tap_kernel<true, 1, 8, 2, 2, 27, 4, 3, false>), grid, dim3(512), 0, s, p);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (p.fuse2) {          // back-to-back 1x1: the 2-D 128 x 256 tile (residual block's 3x3) or the wide stride-2 tile (no residual), lean epilogue
        if ((variant != 4 && variant != 13) || !p.fast_epi || (variant == 4) != (p.has_res != 0) || p.n_tiles_n != 1 || p.Cout != 128 || !p.w2 || !p.b2 || !p.out2 || !p.out2_bytes)
            return hipErrorInvalidValue;
        if (variant == 4) hipLaunchKernelGGL((conv3x3_tap_kernel<false, 2, 4, 4, 4, 27, 4, 2, false, true, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv3x3_tap_kernel<false, 2, 4, 4, 4, 26, 4, 4, false, true, true>), grid, dim3(512), 0, s, p);
        return hipGetLastError();
    }
    switch (variant) {
#define X(id, ...) case id: \
        if (p.f32) hipLaunchKernelGGL((conv3x3_tap_kernel<true, __VA_ARGS__, false>), grid, dim3(512), 0, s, p); \
        else if (p.fast_epi) hipLaunchKernelGGL((conv3x3_tap_kernel<false, __VA_ARGS__, false, true>), grid, dim3(512), 0, s, p); \
        else hipLaunchKernelGGL((conv3x3_tap_kernel<false, __VA_ARGS__, false>), grid, dim3(512), 0, s, p); \
        break;
        YOLO_TAP_VARIANTS(X)
#undef X
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace yolo
