// Shared device pieces of the implicit-GEMM conv kernels (conv.hip: register-staged 4-wave kernel;
// conv_dma.hip: LDS-DMA 8-wave kernel): MFMA wrappers and the fused epilogue.
#pragma once
#include "yolo_internal.h"

namespace yolo {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

#define YOLO_INVALID_OFF 0x80000000u    // >= any buffer size we accept: the range check returns zeros

template <typename T>
__device__ __forceinline__ float4v mma_chunk(const uint4v &a, const uint4v &b, float4v c);

// one 16-byte chunk per lane = 8 halfs: lane group q = lane>>4 holds k = 8q..8q+7 of the 32-deep step
template <>
__device__ __forceinline__ float4v mma_chunk<_Float16>(const uint4v &a, const uint4v &b, float4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}

// fp32 operands: element e of every lane's chunk feeds MFMA e, so the four 16x16x4 steps cover
// k = {e, 4+e, 8+e, 12+e}: all 16 k of the 4 chunks (exact fp32 FMA chain).
template <>
__device__ __forceinline__ float4v mma_chunk<float>(const uint4v &a, const uint4v &b, float4v c) {
    // (element-wise __builtin_bit_cast(float, a.x) miscompiles to element 0 for all four: copy out)
    float af[4], bf[4];
    __builtin_memcpy(af, &a, 16);
    __builtin_memcpy(bf, &b, 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[e], bf[e], c, 0, 0, 0);
    return c;
}

// float32 nets: TWO-LEVEL accumulation.  An f32 MFMA accumulates as one k-ordered fmaf chain, so a 3x3 conv over 1024 input
// channels is a sequential sum of 9216 products: rounding error ~ sqrt(K) ulp of the running sum (measured: 1.3e-4 on
// logits of +-30 through Darknet-19, four times the error of a blocked CPU summation, and above the 1e-4 the fp32 path
// is held to).  The K loop therefore adds its accumulator into a second one every <= 288 k and restarts from zero:
// chain lengths 288 and K / 288 instead of K.
template <int TM, int TP>
__device__ __forceinline__ void flush_acc(float4v (&acc)[TM][TP], float4v (&acc2)[TM][TP]) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) {
            acc2[a][b] += acc[a][b];
            acc[a][b] = float4v{0.f, 0.f, 0.f, 0.f};
        }
}

// Epilogue of one wave: acc[a][b] is the 16x16 tile (cout tile a, pixel tile b); with the
// cout <-> LDS-row permutation of the staging code a lane owns CH = 4*TM CONTIGUOUS couts
// (cbase ..) of pixel (m_wave + 16 b + fr).  bias (folded BN) -> leaky 0.1 (layers.py:6,51) ->
// + residual (shortcut, layers.py:102: no activation after the add) -> store through the output
// index map (identity / nearest-upsample x2 layers.py:115 / block-major reorg layers.py:92-96).
// PADQ (conv_tap.hip): the pixel index is a position q of the padded-linear grid [n][y <= H][x <= W] (one shared
// zero row / column between image rows and images); pad positions are computed but never stored.
// PADQ 0: dense pixel index; 1: padded-linear position (conv_tap.hip MODE 1); 2: position inside 2-D tiles of
// TH x 16 pixels, TH = 16 or 8 (conv_tap.hip MODE 2: qW = tiles per tile row, qHW = tiles per image, t2_shift = log2(16 TH))
template <int PADQ>
__device__ __forceinline__ bool conv_decode_pixel(const ConvParams &p, int m, int &n, int &rem, int &oy, int &ox) {
    bool ok;
    if (PADQ == 2) {
        const int tile = m >> p.t2_shift, l = m & ((1 << p.t2_shift) - 1);
        n = (int)fdiv((uint32_t)tile, p.dqHW);
        const int r = tile - n * p.qHW;
        const int ty = (int)fdiv((uint32_t)r, p.dqW);
        oy = (ty << (p.t2_shift - 4)) + (l >> 4);
        ox = (r - ty * p.qW) * 16 + (l & 15);
        ok = m < p.Mq && oy < p.Ho && ox < p.Wo;
        if (!ok) { n = 0; oy = 0; ox = 0; }
        rem = oy * p.Wo + ox;
    } else if (PADQ == 1) {
        ok = m < p.Mq;
        const int mm = ok ? m : 0;
        n = (int)fdiv((uint32_t)mm, p.dqHW);
        const int r = mm - n * p.qHW;
        oy = (int)fdiv((uint32_t)r, p.dqW);
        ox = r - oy * p.qW;
        ok = ok && ox != p.Wo && oy != p.Ho;
        rem = oy * p.Wo + ox;
    } else {
        ok = m < p.M;
        const int mm = ok ? m : 0;
        n = (int)fdiv((uint32_t)mm, p.dHoWo);
        rem = mm - n * p.HoWo;
        oy = (int)fdiv((uint32_t)rem, p.dWo);
        ox = rem - oy * p.Wo;
    }
    return ok;
}

// The folded-BN bias as the INITIAL value of the accumulators (the MFMA's C input) instead of an add in the epilogue: its
// global load then waits under the prologue's DMA latency rather than in front of the epilogue, where nothing else of the
// workgroup is in flight (block trace: ~0.7 us of a 3.6 us epilogue).  A lane owns couts cbase .. cbase + 4 TM - 1 of every
// fragment (see conv_epilogue).  Kernels that start from it instantiate the epilogues with BIAS_IN_ACC.
template <int TM, int TP>
__device__ __forceinline__ void conv_init_acc_bias(const ConvParams &p, float4v (&acc)[TM][TP], int cbase) {
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const float4v bv = *reinterpret_cast<const float4v *>(p.bias + cbase + 4 * a);       // bias is padded to 128 couts
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = bv;
    }
}

// PIL (conv_tap.hip, round 5: position-interleaved fragments): fragment b of a wave holds the positions m_wave + TP fr + b instead of
// m_wave + 16 b + fr, so that the three taps of a kernel row share their position fragments (see the kernel).
template <bool PIL, int TP>
__device__ __forceinline__ int frag_pos(int m_wave, int b, int fr) { return PIL ? m_wave + TP * fr + b : m_wave + b * 16 + fr; }

template <typename T, int TM, int TP, int PADQ = 0, bool BIAS_IN_ACC = false, bool PIL = false>
__device__ __forceinline__ void conv_epilogue(const ConvParams &p, float4v (&acc)[TM][TP], int cbase, int m_wave, int fr) {
    constexpr int CH = 4 * TM;
    constexpr int EPC = 16 / (int)sizeof(T);
    if (cbase >= p.Cout) return;
    const int nvalid = p.Cout - cbase < CH ? p.Cout - cbase : CH;
    if constexpr (BIAS_IN_ACC) {
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = acc[a][b][j];
                    acc[a][b][j] = p.leaky ? fmaxf(0.1f * x, x) : x;
                }
    } else {   // bias + activation in place: the bias registers die before the residual chunks arrive
        float bias[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) bias[i] = p.bias[cbase + i];      // bias is padded to 128 couts
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = acc[a][b][j] + bias[4 * a + j];
                    acc[a][b][j] = p.leaky ? fmaxf(0.1f * x, x) : x;
                }
    }
    // (fp16) every residual chunk of the wave is requested before the first one is used: the epilogue was a chain
    // of TP dependent load -> store round trips (~1 us each under load, 3.8 us per workgroup in the block trace of
    // conv_tap.hip) while the accumulators sat idle.  The cheap pixel decode is simply done twice.
    constexpr bool PRELOAD = sizeof(T) == 2;
    uint4v rv[PRELOAD ? TP : 1][PRELOAD ? CH / EPC : 1];
    if (PRELOAD && p.has_res && p.vec_res) {
#pragma unroll
        for (int b = 0; b < TP; ++b) {
            int n, rem, oy, ox;
            const bool ok = conv_decode_pixel<PADQ>(p, frag_pos<PIL, TP>(m_wave, b, fr), n, rem, oy, ox);
            // pad / tail lanes read the first pixel's residual (always in range) and ignore it
            const long long ro = ok ? (long long)n * p.res_img_stride + (long long)rem * p.res_ld : 0;
            const T *rp = reinterpret_cast<const T *>(p.res) + ro + cbase;
#pragma unroll
            for (int q = 0; q < CH / EPC; ++q) rv[PRELOAD ? b : 0][PRELOAD ? q : 0] = *reinterpret_cast<const uint4v *>(rp + q * EPC);
        }
    }

#pragma unroll
    for (int b = 0; b < TP; ++b) {
        int n, rem, oy, ox;
        if (!conv_decode_pixel<PADQ>(p, frag_pos<PIL, TP>(m_wave, b, fr), n, rem, oy, ox)) continue;
        float v[CH];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) v[4 * a + j] = acc[a][b][j];
        if (p.has_res) {
            const T *rp = reinterpret_cast<const T *>(p.res) + (long long)n * p.res_img_stride + (long long)rem * p.res_ld + cbase;
            if (p.vec_res) {
#pragma unroll
                for (int q = 0; q < CH / EPC; ++q) {
                    const uint4v u = PRELOAD ? rv[PRELOAD ? b : 0][PRELOAD ? q : 0] : *reinterpret_cast<const uint4v *>(rp + q * EPC);
                    T t[EPC];
                    __builtin_memcpy(t, &u, 16);
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[q * EPC + e] += (float)t[e];
                }
            } else {
                for (int i = 0; i < nvalid; ++i) v[i] += (float)rp[i];
            }
        }
        long long off[4];
        int npos = 1;
        if (p.outmode == OUT_NORMAL) {
            off[0] = (long long)n * p.out_img_stride + (long long)rem * p.out_ld + cbase;
        } else if (p.outmode == OUT_UP2) {
            const long long W2 = 2LL * p.Wo;
            const long long base = (long long)n * p.out_img_stride + ((2LL * oy) * W2 + 2LL * ox) * p.out_ld + cbase;
            off[0] = base; off[1] = base + p.out_ld; off[2] = base + W2 * p.out_ld; off[3] = base + (W2 + 1) * p.out_ld;
            npos = 4;
        } else {
            const int W2 = p.Wo >> 1;
            off[0] = (long long)n * p.out_img_stride + ((long long)(oy >> 1) * W2 + (ox >> 1)) * p.out_ld
                     + ((oy & 1) * 2 + (ox & 1)) * p.Cout + cbase;
        }
        if (p.out_f32) {
            float *op = reinterpret_cast<float *>(p.out);
            for (int q = 0; q < npos; ++q) {
                if (p.vec_out) {
#pragma unroll
                    for (int i = 0; i < CH / 4; ++i)
                        *reinterpret_cast<float4v *>(op + off[q] + 4 * i) = float4v{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
                } else if (nvalid == CH) {
#pragma unroll
                    for (int i = 0; i < CH; ++i) op[off[q] + i] = v[i];
                } else {
                    for (int i = 0; i < nvalid; ++i) op[off[q] + i] = v[i];
                }
            }
        } else {
            T *op = reinterpret_cast<T *>(p.out);
            T t[CH];
#pragma unroll
            for (int i = 0; i < CH; ++i) t[i] = (T)v[i];
            for (int q = 0; q < npos; ++q) {
                if (p.vec_out) {
#pragma unroll
                    for (int i = 0; i < CH / EPC; ++i) {
                        uint4v u;
                        __builtin_memcpy(&u, t + i * EPC, 16);
                        *reinterpret_cast<uint4v *>(op + off[q] + i * EPC) = u;
                    }
                } else {
                    for (int i = 0; i < nvalid; ++i) op[off[q] + i] = t[i];
                }
            }
        }
    }
}

// The common case of the fp16 nets as a LEAN epilogue (ConvParams.fast_epi, set by the launcher): fp16 output of the normal
// index map, 16-byte aligned views below 2 GiB, Cout a multiple of 16, residual (if any) likewise.  The generic epilogue
// above serves every view and output map through run-time branches: ~1 170 of the ~1 480 vector instructions a wave of the dominant
// tap kernel issued (ISA count, profiles/r04_ablation.md) against 576 MFMAs -- selects on p.leaky, 64-bit address arithmetic per
// fragment and access, a canonicalising v_max in front of every fmaxf, the pixel decode done twice.  Here: buffer addressing (one
// 32-bit offset per fragment, invalid lanes carry an out-of-range offset instead of an exec mask), every residual chunk requested
// before the first is used, leaky ReLU as mul + max (max(s x, x) with s = 0.1 or 1: no select; the conv kernels are compiled
// with -fno-honor-nans, so no canonicalisation), + residual in float32 (a code path of its own, no select), ONE rounding, 16-byte stores.  Same values as the generic epilogue for every finite
// input.  The accumulators start from the bias (conv_init_acc_bias).
template <int TM, int TP, int PADQ, bool RES, bool PIL = false>
__device__ __forceinline__ void conv_epilogue_fast_body(const ConvParams &p, float4v (&acc)[TM][TP], int cbase, int m_wave, int fr) {
    typedef _Float16 T;
    constexpr int CH = 4 * TM, EPC = 8, NQ = CH / EPC;
    typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u4;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(RES ? p.res : (const void *)p.out), 0,
                                                                              RES ? p.res_bytes : 0u, 0x00020000);
    const bool c_ok = cbase < p.Cout;
    uint32_t ooff[TP];
    uint4v rv[RES ? TP : 1][NQ];
#pragma unroll
    for (int b = 0; b < TP; ++b) {
        int n, rem, oy, ox;
        const bool ok = conv_decode_pixel<PADQ>(p, frag_pos<PIL, TP>(m_wave, b, fr), n, rem, oy, ox) && c_ok;
        const uint32_t o = (uint32_t)(((long long)n * p.out_img_stride + (long long)rem * p.out_ld + cbase) * 2);
        ooff[b] = ok ? o : YOLO_INVALID_OFF;
        if constexpr (RES) {
            const uint32_t ro = (uint32_t)(((long long)n * p.res_img_stride + (long long)rem * p.res_ld + cbase) * 2);
            const uint32_t roff = ok ? ro : YOLO_INVALID_OFF;
#pragma unroll
            for (int q = 0; q < NQ; ++q) rv[b][q] = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(rs_res, roff, q * 16, 0));
        }
    }
    const float slope = p.leaky ? 0.1f : 1.0f;
#pragma unroll
    for (int b = 0; b < TP; ++b) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            T t[EPC], r[EPC];
            if constexpr (RES) __builtin_memcpy(r, &rv[b][q], 16);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const int i = q * EPC + e;
                float x = acc[i >> 2][b][i & 3];
                x = fmaxf(slope * x, x);        // (-fno-honor-nans for the conv kernels, csrc/Makefile: no canonicalising v_max in front)
                if constexpr (RES) x += (float)r[e];
                t[e] = (T)x;
            }
            u4 u;
            __builtin_memcpy(&u, t, 16);
            __builtin_amdgcn_raw_buffer_store_b128(u, rs_out, ooff[b], q * 16, 0);
        }
    }
}

template <int TM, int TP, int PADQ, bool PIL = false>
__device__ __forceinline__ void conv_epilogue_fast(const ConvParams &p, float4v (&acc)[TM][TP], int cbase, int m_wave, int fr) {
    if (p.has_res) conv_epilogue_fast_body<TM, TP, PADQ, true, PIL>(p, acc, cbase, m_wave, fr);
    else conv_epilogue_fast_body<TM, TP, PADQ, false, PIL>(p, acc, cbase, m_wave, fr);
}

// BACK-TO-BACK 1x1 (ConvParams.fuse2): the 1x1 conv that reads this conv's output (Darknet-53: the first layer of the next
// residual block, net/v3.py:16-19) computed by the SAME workgroup, when its tile holds ALL 128 output channels of its 256 positions
// (8 waves as 2 x 4, TM = TP = 4) and the 1x1 has 64 filters: t[64][256] = W2[64][128] . y[128][256].  The 1x1 launches at 152 x 152
// are HBM-bound re-reads of what the conv in front of them has just written (189 MB read + 95 MB written in 65 us each at batch 32);
// here y goes to HBM as before (the shortcut two layers on and the next 3x3 need it) and, in fp16 exactly as stored, into LDS -- the
// K loop's rings are dead -- as a [256 positions][128 channels] image, chunk c of row r at c ^ (r & 15): conflict-free for the
// ds_write_b128 of the epilogue (8 consecutive rows per write group) and for the ds_read_b128 of the second pass (16 rows of one
// chunk column per lane group).  Second pass: wave w owns positions 32 w .. 32 w + 31 and all 64 filters; A operand = W2 from an LDS
// image behind the first (LDS-DMA, once per workgroup), row of tile a for lane row m = filter 16 (m >> 2) + 4 a + (m & 3), so
// that a lane ends up with 16 CONTIGUOUS filters of its position (32-byte stores); bias as the accumulators' initial value;
// leaky; one rounding.  Same arithmetic as the stand-alone launch up to the
// K order of the 128-deep sum (fp32 accumulation either way).
constexpr int kFuse2ImageBytes = 256 * 256;        // [256 positions][128 channels] fp16
constexpr int kFuse2LdsBytes = kFuse2ImageBytes + 64 * 256;       // + W2 [64 filters][128 channels]: 80 KiB, two workgroups per CU

typedef __attribute__((address_space(3))) void conv_lds_void;
__device__ __forceinline__ void conv_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned char *lds_dst, uint32_t voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (conv_lds_void *)lds_dst, 16, voff, 0, 0, 0);
#else
    (void)rsrc; (void)lds_dst; (void)voff;
#endif
}

template <int PADQ, bool RES>
__device__ __forceinline__ void conv_epilogue_fused_1x1_body(const ConvParams &p, float4v (&acc)[4][4], int q0, int wm, int wn, int wave, int lane,
                                                             unsigned char *smem) {
    typedef _Float16 T;
    constexpr int TP = 4, CH = 16, EPC = 8, NQ = 2;
    typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u4;
    const int fr = lane & 15, fq = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(RES ? p.res : (const void *)p.out), 0,
                                                                              RES ? p.res_bytes : 0u, 0x00020000);
    const int cbase = wm * 64 + fq * CH;            // (n0 = 0: the tile holds every cout)
    const int m_wave = q0 + wn * (TP * 16);
    // ---- first pass epilogue: y -> HBM and -> LDS ----------------------------------------------------------------------------------
    __syncthreads();                                // every wave has left the K loop: the rings are free
    // W2 (64 filters x 128 channels = 16 KiB) comes into LDS ONCE per workgroup, behind the image, by LDS-DMA -- two wave
    // instructions per wave, in flight under the whole first-pass epilogue.  (Read per wave straight from L2 into registers -- 16 KiB
    // x 8 waves per workgroup -- it cost 44 us per launch, as much as the 1x1 launch it replaces: profiles/r04_ablation.md.)
    // Row r = filter r, 16 chunks of 16 bytes; chunk c of row r at c ^ (4 (r >> 4) + (r & 3)): the 16 rows of an A fragment
    // (filters 16 g + 4 a + j for lane row 4 g + j) land on 16 different chunk columns.
    {
        const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w2), 0, p.w2_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = 2 * wave + j;             // 1 KiB piece: rows 4 i .. 4 i + 3
            const int row = 4 * i + (lane >> 4);
            const int logical = (lane & 15) ^ (((row >> 4) << 2) | (row & 3));
            conv_dma16(rs_w2, smem + kFuse2ImageBytes + i * 1024, (uint32_t)row * p.wrow2_bytes + (uint32_t)(logical << 4));
        }
    }
    uint32_t ooff[TP];
    uint4v rv[RES ? TP : 1][NQ];
#pragma unroll
    for (int b = 0; b < TP; ++b) {
        int n, rem, oy, ox;
        const bool ok = conv_decode_pixel<PADQ>(p, m_wave + b * 16 + fr, n, rem, oy, ox);
        const uint32_t o = (uint32_t)(((long long)n * p.out_img_stride + (long long)rem * p.out_ld + cbase) * 2);
        ooff[b] = ok ? o : YOLO_INVALID_OFF;
        if constexpr (RES) {
            const uint32_t ro = (uint32_t)(((long long)n * p.res_img_stride + (long long)rem * p.res_ld + cbase) * 2);
            const uint32_t roff = ok ? ro : YOLO_INVALID_OFF;
#pragma unroll
            for (int q = 0; q < NQ; ++q) rv[b][q] = __builtin_bit_cast(uint4v, __builtin_amdgcn_raw_buffer_load_b128(rs_res, roff, q * 16, 0));
        }
    }
    // (y goes to LDS only here; its global stores come LAST, re-read from the image: `vmcnt` retires in order, and the W2 loads of the
    // second pass queued behind eight HBM stores per lane cost every workgroup ~7 us -- the first build of this fusion was no
    // faster than the two launches)
    const float slope = p.leaky ? 0.1f : 1.0f;
#pragma unroll
    for (int b = 0; b < TP; ++b) {
        const int r = wn * 64 + b * 16 + fr;        // row of the LDS image (r & 15 == fr)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            T t[EPC], rr[EPC];
            if constexpr (RES) __builtin_memcpy(rr, &rv[b][q], 16);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const int i = q * EPC + e;
                float x = acc[i >> 2][b][i & 3];
                x = fmaxf(slope * x, x);
                if constexpr (RES) x += (float)rr[e];
                t[e] = (T)x;
            }
            u4 u;
            __builtin_memcpy(&u, t, 16);
            const int c = wm * 8 + fq * 2 + q;      // 16-byte chunk of the row: channels 8 c .. 8 c + 7
            *reinterpret_cast<u4 *>(smem + r * 256 + ((c ^ fr) << 4)) = u;
        }
    }
    // ---- second pass ------------------------------------------------------------------------------------------------------------------
    float4v acc2[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float4v bv = *reinterpret_cast<const float4v *>(p.b2 + fq * 16 + 4 * a);
        acc2[a][0] = bv; acc2[a][1] = bv;
    }
    __syncthreads();                                // the image is complete
    const unsigned char *Y = smem + (wave * 32 + fr) * 256;
    const unsigned char *W2 = smem + kFuse2ImageBytes + ((fr >> 2) * 16 + (fr & 3)) * 256;      // + 4 a rows: filter 16 (fr >> 2) + 4 a + (fr & 3)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int col = ((4 * ks + fq) ^ fr) << 4;  // (both images swizzle a fragment row by its lane row fr)
        uint4v yb[2], wa[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) wa[a] = *reinterpret_cast<const uint4v *>(W2 + a * 4 * 256 + col);
#pragma unroll
        for (int bt = 0; bt < 2; ++bt) yb[bt] = *reinterpret_cast<const uint4v *>(Y + bt * 16 * 256 + col);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int bt = 0; bt < 2; ++bt) acc2[a][bt] = mma_chunk<T>(wa[a], yb[bt], acc2[a][bt]);
    }
    const __amdgpu_buffer_rsrc_t rs_out2 = __builtin_amdgcn_make_buffer_rsrc(p.out2, 0, p.out2_bytes, 0x00020000);
    const float slope2 = p.leaky2 ? 0.1f : 1.0f;
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
        int n, rem, oy, ox;
        const bool ok = conv_decode_pixel<PADQ>(p, q0 + wave * 32 + bt * 16 + fr, n, rem, oy, ox);
        const uint32_t o = ok ? (uint32_t)(((long long)n * p.out2_img_stride + (long long)rem * p.out2_ld + fq * 16) * 2) : YOLO_INVALID_OFF;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            T t[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const int i = q * EPC + e;          // filter 16 fq + i = tile (i >> 2), row 4 fq + (i & 3)
                const float x = acc2[i >> 2][bt][i & 3];
                t[e] = (T)fmaxf(slope2 * x, x);
            }
            u4 u;
            __builtin_memcpy(&u, t, 16);
            __builtin_amdgcn_raw_buffer_store_b128(u, rs_out2, o, q * 16, 0);
        }
    }
    // y -> HBM: every lane re-reads the chunks it wrote itself (no barrier needed)
#pragma unroll
    for (int b = 0; b < TP; ++b) {
        const int r = wn * 64 + b * 16 + fr;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int c = wm * 8 + fq * 2 + q;
            const u4 u = *reinterpret_cast<const u4 *>(smem + r * 256 + ((c ^ fr) << 4));
            __builtin_amdgcn_raw_buffer_store_b128(u, rs_out, ooff[b], q * 16, 0);
        }
    }
}

// RES is a compile-time property of the instantiation, not a run-time branch: with both bodies in one kernel the register
// allocator spilled 40 registers of the second pass (each body alone: none).  Darknet-53 has exactly the two cases built:
// the residual block's 3x3 (tap kernel, RES) and the stride-2 conv into the stage (LDS-DMA kernel, no residual).
template <int PADQ, bool RES>
__device__ __forceinline__ void conv_epilogue_fused_1x1(const ConvParams &p, float4v (&acc)[4][4], int q0, int wm, int wn, int wave, int lane,
                                                        unsigned char *smem) {
    conv_epilogue_fused_1x1_body<PADQ, RES>(p, acc, q0, wm, wn, wave, lane, smem);
}

// conv + 2x2/2 max-pool (net/layers.py:70-81 behind net/layers.py:17-67; even H and W, so the pool's zero pad row / column is never
// read) for the 2-D tiles of conv_tap.hip (PADQ 2): fragment b of a wave is tile row (first row of the wave) + b, TP is even and
// tiles start on even rows, so the two rows of a pool window are fragments b, b + 1 of the SAME lane and its two columns are lanes
// fr, fr ^ 1.  Activation first (the reference pools the activated tensor), maximum in float32, one rounding, even lanes store the
// pooled pixel (oy / 2, ox / 2) of a [Ho / 2][Wo / 2] tensor.  No residual (a conv that feeds a pool has none in this vocabulary).
template <typename T, int TM, int TP>
__device__ __forceinline__ void conv_epilogue_pool2(const ConvParams &p, float4v (&acc)[TM][TP], int cbase, int m_wave, int fr) {
    constexpr int CH = 4 * TM;
    constexpr int EPC = 16 / (int)sizeof(T);
    static_assert(TP % 2 == 0, "pool windows pair the fragments of a wave");
    if (cbase >= p.Cout) return;
#pragma unroll
    for (int b = 0; b < TP; b += 2) {
        int n, rem, oy, ox;
        const bool ok = conv_decode_pixel<2>(p, m_wave + b * 16 + fr, n, rem, oy, ox);       // row oy even; oy + 1 < Ho as Ho is even
        float v[CH];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x0 = acc[a][b][j], x1 = acc[a][b + 1][j];
                if (p.leaky) { x0 = fmaxf(0.1f * x0, x0); x1 = fmaxf(0.1f * x1, x1); }
                const float m = fmaxf(x0, x1);
                v[4 * a + j] = fmaxf(m, __shfl_xor(m, 1));          // lanes fr, fr ^ 1: columns ox, ox ^ 1 (same lane group fq)
            }
        if (!ok || (fr & 1)) continue;
        const long long off = (long long)n * p.out_img_stride + ((long long)(oy >> 1) * (p.Wo >> 1) + (ox >> 1)) * p.out_ld + cbase;
        T *op = reinterpret_cast<T *>(p.out);
        T t[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) t[i] = (T)v[i];
#pragma unroll
        for (int i = 0; i < CH / EPC; ++i) {
            uint4v u;
            __builtin_memcpy(&u, t + i * EPC, 16);
            *reinterpret_cast<uint4v *>(op + off + i * EPC) = u;
        }
    }
}

// Split-K: the wave's raw accumulators (no bias, no activation) go to the float32 slab part[split][pixel][cout_pad]; a lane
// owns CH contiguous couts of a pixel -> 16-byte stores, 64 contiguous bytes per lane and fragment.
template <int TM, int TP, int PADQ = 0, bool PIL = false>
__device__ __forceinline__ void conv_store_partial(const ConvParams &p, float4v (&acc)[TM][TP], int cbase, int m_wave, int fr, int split) {
    if (cbase >= p.cout_pad) return;
#pragma unroll
    for (int b = 0; b < TP; ++b) {
        int n, rem, oy, ox;
        if (!conv_decode_pixel<PADQ>(p, frag_pos<PIL, TP>(m_wave, b, fr), n, rem, oy, ox)) continue;
        float *dst = p.part + ((size_t)split * (size_t)p.M + (size_t)(n * p.HoWo + rem)) * (size_t)p.cout_pad + cbase;
#pragma unroll
        for (int a = 0; a < TM; ++a) *reinterpret_cast<float4v *>(dst + 4 * a) = acc[a][b];
    }
}

// Head convs: float32 output with a channel count that is not a multiple of 4 (255 = 3 x 85, 425 = 5 x 85), so the
// generic epilogue falls back to one scattered 4-byte store per value (64 lanes -> 64 different 4-byte pieces per
// instruction).  Here each wave transposes its 16 pixels x (4 CH) couts through a private LDS slab and writes every
// pixel's run of 64 consecutive floats (256 contiguous bytes) with one instruction.  OUT_NORMAL, no residual.
constexpr int kStagePitch(int TM) { return 16 * TM + 4; }       // floats per pixel row of the slab (4 CH + pad)

// SPARSE ROWS (detect only: ConvParams.obj_min > -inf; `flags` = workgroup-shared LDS, [pixels of the tile][8 anchors], pix0 = the
// wave's first pixel in it): a (cell, anchor) row of 5 + classes logits whose objectness logit is below obj_min can never pass the
// score threshold (v3: p = sigmoid(obj); v2: p = sigmoid(obj) * softmax <= sigmoid(obj); obj_min = logit(threshold) - 0.01), and
// the decode kernel reads the rows of passing candidates only (detect.hip), so such rows are not written at all: the compact
// objectness array stays complete, the 76 x 76 head of YOLOv3-608 stops writing 188 MB of float32 per batch of 32.  The lanes that
// hold an objectness channel publish the verdict per (pixel, anchor); after a workgroup barrier every wave skips the stores of the
// columns of a failed anchor.  yolo_net_forward (dense logits for the caller) never sets obj_min.
constexpr int kStageFlagAnchors = 8;

// NBW: pixels of the WORKGROUP's tile (rows of `flags`)
template <int TM, int TP, int PADQ = 0, bool BIAS_IN_ACC = false, int NBW = 0>
__device__ __forceinline__ void conv_epilogue_f32_staged(const ConvParams &p, float4v (&acc)[TM][TP], int cbase_wave, int m_wave,
                                                         int lane, float *slab, float *flags = nullptr, int pix0 = 0) {
    constexpr int CH = 4 * TM;
    constexpr int PITCH = kStagePitch(TM);
    const int fr = lane & 15, fq = lane >> 4;
    const int cbase = cbase_wave + fq * CH;
    if constexpr (BIAS_IN_ACC) {
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = acc[a][b][j];
                    acc[a][b][j] = p.leaky ? fmaxf(0.1f * x, x) : x;
                }
    } else {
        float bias[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) bias[i] = p.bias[cbase + i];      // bias is padded to 128 couts
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = acc[a][b][j] + bias[4 * a + j];
                    acc[a][b][j] = p.leaky ? fmaxf(0.1f * x, x) : x;
                }
    }
    float *op = reinterpret_cast<float *>(p.out);
    // objectness logits (channel a * (5+C) + 4 of anchor a) also go to a compact [B][rows] array for the decode kernel,
    // which otherwise touches one 64-byte sector per row to read 4 bytes of it
    constexpr int NH = (4 * CH + 63) / 64;
    int obj_a[NH], col_a[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        const int cg = cbase_wave + lane + 64 * h;
        const int a = p.obj_out ? cg / p.obj_width : 0;
        obj_a[h] = (p.obj_out && lane + 64 * h < 4 * CH && cg < p.Cout && cg - a * p.obj_width == 4) ? a : -1;
        col_a[h] = a < kStageFlagAnchors ? a : kStageFlagAnchors - 1;
    }
    const bool sparse = NBW > 0 && flags != nullptr && p.obj_out != nullptr && p.obj_min > -3.0e38f && p.obj_na <= kStageFlagAnchors &&
                        p.obj_width > CH;      // kernel-uniform; (5 + classes > CH: a lane's CH couts hold at most ONE objectness channel)
    float objv[TP];             // the objectness logits this lane holds (pixel fr of every fragment), anchor obj_an; -1: none
    int obj_an = -1;
    if (sparse) {
        // The verdict of an anchor exists only in the workgroup whose cout tile holds that anchor's objectness channel (255 or 425 head
        // channels span two to four 128-cout tiles: anchor 1's channel 89 sits in tile 0, its class columns 128..169 in tile 1).  Every
        // flag therefore starts as "write the row": the columns of an anchor this workgroup cannot judge are always stored, and only the
        // lanes that own an objectness channel may turn a flag off.  (Round 4 read those slots uninitialised -- leftover ring bytes.)
        {
            const int n_flags = NBW * kStageFlagAnchors;
            for (int i = (int)threadIdx.x; i < n_flags; i += (int)blockDim.x) flags[i] = 1.f;
            __syncthreads();
        }
        int r = cbase % p.obj_width, a = cbase / p.obj_width;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (r == 4 && cbase + i < p.Cout) {
                obj_an = a;
#pragma unroll
                for (int b = 0; b < TP; ++b) {
                    objv[b] = acc[i >> 2][b][i & 3];
                    flags[(pix0 + b * 16 + fr) * kStageFlagAnchors + a] = objv[b] >= p.obj_min ? 1.f : 0.f;
                }
            }
            if (++r == p.obj_width) { r = 0; ++a; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < TP; ++b) {
        int n, rem, oy, ox;
        const bool ok = conv_decode_pixel<PADQ>(p, m_wave + b * 16 + fr, n, rem, oy, ox);
        const long long off = (long long)n * p.out_img_stride + (long long)rem * p.out_ld;
        const int off_lo = (int)(off & 0xffffffffLL), off_hi = (int)(off >> 32), oki = ok ? 1 : 0;
        const int orow = n * p.obj_rows + p.obj_row0 + rem * p.obj_na;
        if (sparse) {
            // no anchor of these 16 pixels can be a candidate (the usual case): nothing of the fragment goes to the logits, its
            // objectness logits go to the compact array straight from the lanes that hold them -- no slab, no store loop
            // (only the anchors that own one of THIS wave's columns count: the others' rows are stored by other waves / workgroups)
            bool any = false;
            if (lane < 16) {
                const int c_end = (cbase_wave + 4 * CH < p.Cout ? cbase_wave + 4 * CH : p.Cout) - 1;
                const int a_lo = cbase_wave / p.obj_width, a_hi = c_end / p.obj_width < p.obj_na ? c_end / p.obj_width : p.obj_na - 1;
                for (int a = a_lo; a <= a_hi; ++a) any = any || flags[(pix0 + b * 16 + lane) * kStageFlagAnchors + a] != 0.f;
            }
            if (!__builtin_amdgcn_ballot_w64(any)) {
                if (ok && obj_an >= 0) p.obj_out[orow + obj_an] = objv[b];
                continue;
            }
        }
#pragma unroll
        for (int a = 0; a < TM; ++a) *reinterpret_cast<float4v *>(slab + fr * PITCH + fq * CH + 4 * a) = acc[a][b];
        __builtin_amdgcn_wave_barrier();        // LDS executes a wave's instructions in order: the reads below see these writes
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) {
            if (!__builtin_amdgcn_readlane(oki, pp)) continue;         // wave-uniform
            const long long po = ((long long)__builtin_amdgcn_readlane(off_hi, pp) << 32) | (unsigned)__builtin_amdgcn_readlane(off_lo, pp);
            const int orow_pp = __builtin_amdgcn_readlane(orow, pp);
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const int c = lane + 64 * h;
                if (c < 4 * CH && cbase_wave + c < p.Cout) {
                    const float v = slab[pp * PITCH + c];
                    if (!sparse || flags[(pix0 + b * 16 + pp) * kStageFlagAnchors + col_a[h]] != 0.f) op[po + cbase_wave + c] = v;
                    if (obj_a[h] >= 0) p.obj_out[orow_pp + obj_a[h]] = v;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// XCD-aware bijective remap of the linear block id: the 8 XCDs (blocks b, b+8, ... share one) get
// contiguous ranges of (pixel-tile, cout-tile) so the cout tiles of one pixel tile run back to back
// on one XCD and share its L2.  Speed only: any placement gives the same result.
__device__ __forceinline__ int xcd_remap(int bid, int n_blocks) {
    const int q = n_blocks >> 3, r = n_blocks & 7, x = bid & 7, y = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
}

}  // namespace yolo
