// Fused conv2d + folded-BN bias + leaky ReLU [+ residual] [+ upsample / reorg / concat-slice write]
// as an implicit GEMM on the gfx950 matrix cores.  Replaces tf.layers.conv2d +
// tf.layers.batch_normalization + tf.nn.leaky_relu (net/layers.py:17-67), and absorbs
// shortcut (:100-103), route (:84-87), upsample (:112-116) and reorg (:90-97).
//
// GEMM view:  D[cout][pixel] = sum_k  Wt[cout][k] * X[pixel][k],   k = (kh, kw, cin)
//   * the WEIGHT tile is the MFMA A operand and the PIXEL tile the B operand, so that in the
//     16x16 accumulator a lane owns 4 consecutive couts of ONE pixel; with the cout<->LDS-row
//     permutation below a lane ends up with 4*TM contiguous channels of a pixel and the
//     epilogue writes 16/32-byte NHWC vectors (four lanes = one 64/128-byte line).
//   * K runs in 16-byte chunks (8 halfs / 4 floats).  A K tile is 8 chunks = one 128-byte LDS
//     row per cout / pixel.  "uniform" mode: Cin is a multiple of 8 chunks, the tap (kh,kw) is
//     the same for the whole tile.  "perchunk" mode (Cin = 1, 2 or 4 chunks: first layer and the
//     narrow early layers): every chunk of the tile carries its own tap.
//   * im2col is never materialised: each thread keeps, per pixel row it stages, a 32-bit byte
//     offset and a 9-bit tap-validity mask; padding (layers.py:9-14) and the M tail are served
//     by the buffer descriptor's range check (offset 0x80000000 reads zeros).
//   * LDS image: row-major 128-byte rows, chunk index XOR ((row>>1)&7): conflict-free for the
//     ds_read_b128 fragment reads (16 rows x one chunk per 16-lane group) and the ds_write_b128
//     staging writes.
//   * pipeline: global loads of tile t+1 are issued before the MFMAs of tile t and written to the
//     other LDS buffer after them; one barrier per K tile.
#include "conv_common.h"
#include <type_traits>

namespace yolo {

#define INVALID_OFF YOLO_INVALID_OFF

// WM x WN waves; a wave owns TM*16 couts x TP*16 pixels.
// (F32 rather than the element type as template parameter: see conv_tap.hip)
template <bool F32, int WM, int WN, int TM, int TP, bool PERCHUNK>
__global__ void __launch_bounds__(256, 2) conv_igemm_kernel(const ConvParams p) {      // two workgroups per CU: <= 256 registers
    typedef typename std::conditional<F32, float, _Float16>::type T;
    static_assert(WM * WN == 4, "four waves per workgroup");
    constexpr int NA = WM * TM * 16;        // couts per block
    constexpr int NB = WN * TP * 16;        // pixels per block
    constexpr int LA = NA / 32;             // 16-byte staging loads per thread per tile
    constexpr int LB = NB / 32;
    constexpr int CH = 4 * TM;              // contiguous channels a lane owns per pixel
    constexpr int ES = (int)sizeof(T);
    constexpr int TILE_BYTES = (NA + NB) * 128;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int bid = xcd_remap(blockIdx.x, p.n_blocks);
    const int mt = (int)fdiv((uint32_t)bid, p.dtiles_n);
    const int nt = bid - mt * p.n_tiles_n;
    const int n0 = nt * NA;
    const int m0 = mt * NB;

    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wgt), 0, p.wgt_bytes, 0x00020000);

    // ---- staging geometry: thread -> (row r0 + 32 i, chunk c) of each tile ------------------
    const int c = tid & 7;
    const int r0 = tid >> 3;
    const int sw = (c ^ ((r0 >> 1) & 7)) << 4;      // swizzled chunk byte offset (same for every i)

    uint32_t a_off[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int r = r0 + 32 * i;
        const int ws = r / (TM * 16), R = r % (TM * 16);
        const int tm = R >> 4, g = (R >> 2) & 3, j = R & 3;
        const int ch = ws * (TM * 16) + g * CH + 4 * tm + j;    // LDS row R holds this cout
        a_off[i] = (uint32_t)(n0 + ch) * p.wrow_bytes + c * 16;
    }
    uint32_t b_base[LB], b_mask[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int m = m0 + r0 + 32 * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = (int)fdiv((uint32_t)mm, p.dHoWo);
        const int rem = mm - n * p.HoWo;
        const int oy = (int)fdiv((uint32_t)rem, p.dWo);
        const int ox = rem - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        const long long e = (long long)n * p.in_img_stride + ((long long)iy0 * p.W + ix0) * p.in_ld + p.in_coff;
        b_base[i] = (uint32_t)(e * ES) + (PERCHUNK ? 0 : c * 16);
        uint32_t mask = 0;
        if (ok) {
            for (int t = 0; t < p.taps; ++t) {
                const int kh = p.ksize == 3 ? (t * 11) >> 5 : 0, kw = t - kh * p.ksize;    // ksize is 1 or 3
                if ((unsigned)(iy0 + kh) < (unsigned)p.H && (unsigned)(ix0 + kw) < (unsigned)p.W) mask |= 1u << t;
            }
        }
        b_mask[i] = mask;
    }

    uint4v ra[LA], rb[LB];

    auto load_tile = [&](int kt) {
        const uint32_t ka = (uint32_t)kt * 128;
#pragma unroll
        for (int i = 0; i < LA; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_off[i] + ka, 0, 0);
        int tap;
        uint32_t koff;
        if (PERCHUNK) {
            const int kc = kt * 8 + c;
            tap = kc >> p.cpt_shift;
            koff = (uint32_t)(kc & ((1 << p.cpt_shift) - 1)) * 16;
        } else {
            tap = (int)fdiv((uint32_t)kt, p.dtpt);
            koff = (uint32_t)(kt - tap * p.tiles_per_tap) * 128;
        }
        const int kh = p.ksize == 3 ? (tap * 11) >> 5 : 0;
        const int kw = tap - kh * p.ksize;
        const uint32_t toff = (uint32_t)((kh * p.W + kw) * p.in_ld * ES) + koff;
        const bool tap_ok = tap < p.taps;
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const bool ok = tap_ok && ((b_mask[i] >> (tap & 15)) & 1u);
            rb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? b_base[i] + toff : INVALID_OFF, 0, 0);
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char *A = smem + buf * TILE_BYTES;
        unsigned char *B = A + NA * 128;
#pragma unroll
        for (int i = 0; i < LA; ++i) *reinterpret_cast<uint4v *>(A + (r0 + 32 * i) * 128 + sw) = ra[i];
#pragma unroll
        for (int i = 0; i < LB; ++i) *reinterpret_cast<uint4v *>(B + (r0 + 32 * i) * 128 + sw) = rb[i];
    };

    float4v acc[TM][TP];
    float4v acc2[F32 ? TM : 1][F32 ? TP : 1];       // float32: second-level accumulator (conv_common.h: flush_acc)
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) {
            acc[a][b] = float4v{0.f, 0.f, 0.f, 0.f};
            if (F32) acc2[F32 ? a : 0][F32 ? b : 0] = float4v{0.f, 0.f, 0.f, 0.f};
        }

    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int buf) {
        const unsigned char *A = smem + buf * TILE_BYTES + (wm * TM * 16 + fr) * 128;
        const unsigned char *B = smem + buf * TILE_BYTES + NA * 128 + (wn * TP * 16 + fr) * 128;
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

    // ---- main loop ------------------------------------------------------------------------
    // K tiles of this workgroup: all of them, or one K split's share (split-K, blockIdx.y)
    const int kt0 = p.ksplit > 1 ? (int)blockIdx.y * p.kunits : 0;
    const int kt1 = p.ksplit > 1 ? (kt0 + p.kunits < p.ktiles ? kt0 + p.kunits : p.ktiles) : p.ktiles;
    load_tile(kt0);
    store_tile(0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        const bool more = kt + 1 < kt1;
        if (more) load_tile(kt + 1);
        compute(cur);
        if constexpr (F32) {            // a K tile is 32 floats deep: restart the chain every 256 k
            if (((kt - kt0) & 7) == 7 || !more) flush_acc<TM, TP>(acc, acc2);
        }
        if (more) store_tile(cur ^ 1);
        __syncthreads();
    }
    if constexpr (F32) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TP; ++b) acc[a][b] = acc2[a][b];
    }

    if (p.ksplit > 1) {     // split-K: raw accumulators to the float32 slab, epilogue in splitk_reduce_kernel
        conv_store_partial<TM, TP>(p, acc, n0 + wm * (TM * 16) + fq * CH, m0 + wn * (TP * 16), fr, (int)blockIdx.y);
        return;
    }
    // ---- epilogue: bias, leaky, residual, store (conv_common.h) -------------------------------
    if (p.out_f32 && !p.vec_out && p.outmode == OUT_NORMAL && !p.has_res) {      // head conv: coalesced float32 rows via LDS
        static_assert(4 * 16 * kStagePitch(TM) * 4 <= 2 * TILE_BYTES, "staging slabs must fit in the tile buffers");
        // (the loop's last __syncthreads already separates the tile reads from this reuse)
        conv_epilogue_f32_staged<TM, TP>(p, acc, n0 + wm * (TM * 16), m0 + wn * (TP * 16), lane,
                                         reinterpret_cast<float *>(smem) + wave * 16 * kStagePitch(TM));
        return;
    }
    conv_epilogue<T, TM, TP>(p, acc, n0 + wm * (TM * 16) + fq * CH, m0 + wn * (TP * 16), fr);
}

// ---- float32 convs on the bf16 matrix cores: every product as nine bf16 x bf16 products (yolo_net_options.f32_products; round 5: by rule) ----------
// An fp32 value is EXACTLY the sum of three bf16 values (8 + 8 + 8 mantissa bits, same exponent range): x = h + m + l with
// h = bf16(x), m = bf16(x - h), l = bf16(x - h - m).  Then a b = sum over the nine (i, j) of a_i b_j, each an exact product of two
// 8-bit mantissas accumulated in fp32 by mfma_f32_16x16x32_bf16 -- at 16x the rate of mfma_f32_16x16x4f32, i.e. 16 / 9 of the float32
// matrix peak.  Operands are split while they are staged (global fp32 -> registers -> three bf16 planes in LDS); the small terms are
// added first; the accumulator is flushed into a second one every 256 k as in the fp32 kernel.  Measured (tiny-YOLOv2-VOC b64, the two
// 13 x 13 layers that are 65 % of its step): logits 2.0e-5 from the CPU oracle (native fp32 MFMA: 2.2e-5), boxes identical;
// 1024 -> 1024 1.81 -> 1.59 ms (127 TFLOP/s against the 137 a register-fed mfma_f32_16x16x4f32 loop sustains), step 15.3 k -> 16.7 k
// img/s.  Round 5: taken BY RULE (conv_f32_emu_rule below) by the long-K whole-K launches of float32 nets -- exact partial products, float32
// accumulation: the 1e-4 logit contract of the float32 path is asserted with it (tests/test_gpu_nets.py) -- and `f32_products = 1` keeps the
// native float32 MFMA everywhere (the A/B arm, and what profiles/ up to round 4 measured).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

template <int WM, int WN, int TM, int TP>
__global__ void __launch_bounds__(512) conv_igemm_emu_kernel(const ConvParams p) {
    typedef float T;
    static_assert(WM * WN == 8, "eight waves per workgroup, one workgroup per CU (96 KiB of LDS): two waves per SIMD, so that one wave's split + staging runs under the other's MFMAs");
    constexpr int NA = WM * TM * 16, NB = WN * TP * 16, LA = NA / 64, LB = NB / 64, CH = 4 * TM;
    constexpr int PA = NA * 64, PB = NB * 64;               // one bf16 plane of the weight / pixel tile: 64-byte rows (32 k)
    constexpr int TILE_BYTES = 3 * (PA + PB);
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int bid = xcd_remap(blockIdx.x, p.n_blocks);
    const int mt = (int)fdiv((uint32_t)bid, p.dtiles_n);
    const int nt = bid - mt * p.n_tiles_n;
    const int n0 = nt * NA, m0 = mt * NB;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.wgt), 0, p.wgt_bytes, 0x00020000);
    const int c = tid & 7, r0 = tid >> 3;
    uint32_t a_off[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int r = r0 + 64 * i;
        const int ws = r / (TM * 16), R = r % (TM * 16);
        const int tm = R >> 4, g = (R >> 2) & 3, j = R & 3;
        a_off[i] = (uint32_t)(n0 + ws * (TM * 16) + g * CH + 4 * tm + j) * p.wrow_bytes + c * 16;
    }
    uint32_t b_base[LB], b_mask[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int m = m0 + r0 + 64 * i;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int n = (int)fdiv((uint32_t)mm, p.dHoWo);
        const int rem = mm - n * p.HoWo;
        const int oy = (int)fdiv((uint32_t)rem, p.dWo), ox = rem - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        const long long e = (long long)n * p.in_img_stride + ((long long)iy0 * p.W + ix0) * p.in_ld + p.in_coff;
        b_base[i] = (uint32_t)(e * 4) + c * 16;
        uint32_t mask = 0;
        if (ok)
            for (int t = 0; t < p.taps; ++t) {
                const int kh = p.ksize == 3 ? (t * 11) >> 5 : 0, kw = t - kh * p.ksize;
                if ((unsigned)(iy0 + kh) < (unsigned)p.H && (unsigned)(ix0 + kw) < (unsigned)p.W) mask |= 1u << t;
            }
        b_mask[i] = mask;
    }
    // two register sets: the loads of tile kt + 2 are in flight while tile kt + 1 is split under the MFMAs of tile kt
    uint4v ra[2][LA], rb[2][LB];
    auto load_tile = [&](int kt, int set) {
        const uint32_t ka = (uint32_t)kt * 128;
#pragma unroll
        for (int i = 0; i < LA; ++i) ra[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, a_off[i] + ka, 0, 0);
        const int tap = (int)fdiv((uint32_t)kt, p.dtpt);
        const uint32_t koff = (uint32_t)(kt - tap * p.tiles_per_tap) * 128;
        const int kh = p.ksize == 3 ? (tap * 11) >> 5 : 0, kw = tap - kh * p.ksize;
        const uint32_t toff = (uint32_t)((kh * p.W + kw) * p.in_ld * 4) + koff;
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const bool ok = tap < p.taps && ((b_mask[i] >> (tap & 15)) & 1u);
            rb[set][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? b_base[i] + toff : INVALID_OFF, 0, 0);
        }
    };
    // four floats -> three planes of four bf16 (8 bytes each) at row r, k = 4 c .. 4 c + 3
    auto put = [&](unsigned char *plane0, int plane_bytes, int r, const uint4v &v) {
        float x[4];
        __builtin_memcpy(x, &v, 16);
        bf16x4_t h, m, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const __bf16 hh = (__bf16)x[e];
            const float r1 = x[e] - (float)hh;
            const __bf16 mm = (__bf16)r1;
            const float r2 = r1 - (float)mm;
            h[e] = hh; m[e] = mm; l[e] = (__bf16)r2;
        }
        unsigned char *dst = plane0 + r * 64 + c * 8;
        *reinterpret_cast<bf16x4_t *>(dst) = h;
        *reinterpret_cast<bf16x4_t *>(dst + plane_bytes) = m;
        *reinterpret_cast<bf16x4_t *>(dst + 2 * plane_bytes) = l;
    };
    auto store_tile = [&](int buf, int set) {
        unsigned char *A = smem + buf * TILE_BYTES, *B = A + 3 * PA;
#pragma unroll
        for (int i = 0; i < LA; ++i) put(A, PA, r0 + 64 * i, ra[set][i]);
#pragma unroll
        for (int i = 0; i < LB; ++i) put(B, PB, r0 + 64 * i, rb[set][i]);
    };
    float4v acc[TM][TP], acc2[TM][TP];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) { acc[a][b] = float4v{0.f, 0.f, 0.f, 0.f}; acc2[a][b] = float4v{0.f, 0.f, 0.f, 0.f}; }
    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int buf) {
        const unsigned char *A = smem + buf * TILE_BYTES + (wm * TM * 16 + fr) * 64 + fq * 16;
        const unsigned char *B = smem + buf * TILE_BYTES + 3 * PA + (wn * TP * 16 + fr) * 64 + fq * 16;
        // the nine plane pairs, small terms first: (l,l) (m,l) (l,m) (h,l) (l,h) (m,m) (h,m) (m,h) (h,h)
        constexpr int PI[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0}, PJ[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            bf16x8_t fa[TM], fb[TP];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = *reinterpret_cast<const bf16x8_t *>(A + PI[t] * PA + a * 16 * 64);
#pragma unroll
            for (int b = 0; b < TP; ++b) fb[b] = *reinterpret_cast<const bf16x8_t *>(B + PJ[t] * PB + b * 16 * 64);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TP; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
    };
    const int kt1 = p.ktiles;
    load_tile(0, 0);
    store_tile(0, 0);
    if (kt1 > 1) load_tile(1, 1);
    __syncthreads();
    // iteration kt: request tile kt + 2 (register set kt & 1), MFMAs of tile kt (LDS buffer kt & 1) interleaved with the split of
    // tile kt + 1 (register set (kt + 1) & 1, requested an iteration ago) into the other LDS buffer
    auto iter = [&](int kt, auto parc) {
        constexpr int par = decltype(parc)::value;
        if (kt + 2 < kt1) load_tile(kt + 2, par);
        compute(par);
        if (kt + 1 < kt1) store_tile(par ^ 1, par ^ 1);
        // (tried on top, both without effect on the 1.59 ms of the 1024 -> 1024 layer: `sched_group_barrier` pairs asking for one MFMA /
        // two vector instructions in turn -- the scheduler still emits the 72 MFMAs first --, and the nine plane pairs with one staged
        // piece's split pinned behind each by scheduling barriers: 1.62 ms.  The split is not what the MFMAs wait for.)
        if ((kt & 7) == 7 || kt + 1 >= kt1) flush_acc<TM, TP>(acc, acc2);
        __syncthreads();
    };
    for (int kt = 0; kt < kt1; kt += 2) {
        iter(kt, std::integral_constant<int, 0>());
        if (kt + 1 < kt1) iter(kt + 1, std::integral_constant<int, 1>());
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) acc[a][b] = acc2[a][b];
    if (p.out_f32 && !p.vec_out && p.outmode == OUT_NORMAL && !p.has_res) {
        static_assert(8 * 16 * kStagePitch(TM) * 4 <= 2 * TILE_BYTES, "staging slabs must fit in the tile buffers");
        conv_epilogue_f32_staged<TM, TP>(p, acc, n0 + wm * (TM * 16), m0 + wn * (TP * 16), lane,
                                         reinterpret_cast<float *>(smem) + wave * 16 * kStagePitch(TM));
        return;
    }
    conv_epilogue<T, TM, TP>(p, acc, n0 + wm * (TM * 16) + fq * CH, m0 + wn * (TP * 16), fr);
}

template <bool F32, bool PC>
static hipError_t launch_cfg(const ConvParams &p0, int cfg, hipStream_t s) {
    ConvParams p = p0;
    int na, nb;
    switch (cfg) {
    case CFG_N128: na = 128; nb = 128; break;
    case CFG_N64: na = 64; nb = 256; break;
    default: na = 32; nb = 256; break;
    }
    p.n_tiles_n = (p.Cout + na - 1) / na;
    const long long mt = ((long long)p.M + nb - 1) / nb;
    const long long blocks = mt * p.n_tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    p.n_blocks = (int)blocks;
    conv_set_divisors(p, p.tiles_per_tap);
    if (p.ksplit > 1 && (!p.part || p.kunits < 1 || (long long)p.ksplit * p.kunits < p.ktiles)) return hipErrorInvalidValue;
    dim3 grid((unsigned)blocks, (unsigned)(p.ksplit > 1 ? p.ksplit : 1)), block(256);
    if (F32 && !PC && p.f32_emu && cfg == CFG_N128 && p.ksplit <= 1) {
        hipLaunchKernelGGL((conv_igemm_emu_kernel<2, 4, 4, 2>), grid, dim3(512), 0, s, p);
        return hipGetLastError();
    }
    switch (cfg) {
    case CFG_N128: hipLaunchKernelGGL((conv_igemm_kernel<F32, 2, 2, 4, 4, PC>), grid, block, 0, s, p); break;
    case CFG_N64: hipLaunchKernelGGL((conv_igemm_kernel<F32, 1, 4, 4, 4, PC>), grid, block, 0, s, p); break;
    default: hipLaunchKernelGGL((conv_igemm_kernel<F32, 1, 4, 2, 4, PC>), grid, block, 0, s, p); break;
    }
    return hipGetLastError();
}

hipError_t launch_conv(const ConvParams &p, int dtype, int cfg, bool perchunk, hipStream_t s) {
    if (dtype == YOLO_DTYPE_F16)
        return perchunk ? launch_cfg<false, true>(p, cfg, s) : launch_cfg<false, false>(p, cfg, s);
    return perchunk ? launch_cfg<true, true>(p, cfg, s) : launch_cfg<true, false>(p, cfg, s);
}

// nine bf16 products instead of the float32 MFMA: the 128 x 128 whole-K launches of float32 nets that are matrix-bound -- K of at least
// 4608 (nine taps x 512 channels) over at least a chip's worth of workgroups -- or, with f32_products = 2, every launch the kernel applies to
bool conv_f32_emu_rule(int f32_products, int dtype, const ConvParams &p, int cfg, bool perchunk, int ksplit) {
    if (dtype != YOLO_DTYPE_F32 || perchunk || cfg != CFG_N128 || ksplit > 1 || f32_products == 1) return false;
    if (f32_products == 2) return true;
    const long long blocks = ((long long)p.M + 127) / 128 * ((p.Cout + 127) / 128);
    return (long long)p.ktiles * 32 >= 4608 && blocks >= 256;
}

// the name rocprofv3's kernel trace prints (yolo_kernel_info.symbol)
std::string conv_symbol(int dtype, int cfg, bool perchunk, bool f32_emu) {
    if (f32_emu) return "void yolo::conv_igemm_emu_kernel<2, 4, 4, 2>(yolo::ConvParams)";
    const char *shape = cfg == CFG_N128 ? "2, 2, 4, 4" : cfg == CFG_N64 ? "1, 4, 4, 4" : "1, 4, 2, 4";
    return std::string("void yolo::conv_igemm_kernel<") + (dtype == YOLO_DTYPE_F16 ? "false" : "true") + ", " + shape + ", " +
           (perchunk ? "true" : "false") + ">(yolo::ConvParams)";
}

}  // namespace yolo
